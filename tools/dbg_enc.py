import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import numpy as np, torch
import super_sac_amd as ssa
import ssac_oracle as orc
import case_runner
from super_sac_amd import conv_encoder
DEV='cuda'
kind, ch, emb = 'big', 9, 50
p = orc.make_conv_encoder(np.random.RandomState(60+ch), kind, ch, emb)
conv = ssa.nets.BigPixelEncoder((ch,84,84), emb)
with torch.no_grad():
    for i, nm in enumerate(["conv1","conv2","conv3","conv4"],1):
        getattr(conv,nm).weight.copy_(p[f"c{i}w"]); getattr(conv,nm).bias.copy_(p[f"c{i}b"])
    conv.fc.weight.copy_(p["fcw"]); conv.fc.bias.copy_(p["fcb"]); conv.ln.weight.copy_(p["lnw"]); conv.ln.bias.copy_(p["lnb"])
conv = conv.to(DEV)
rng = np.random.RandomState(7); B=5
x = torch.from_numpy(rng.randint(0,256,(B,ch,84,84)).astype(np.float32))
d_rep = torch.from_numpy(rng.standard_normal((B,emb)).astype(np.float32))
pr = {k: v.clone().requires_grad_(True) for k,v in p.items()}
y = orc.encode({"kind":kind,"key":"obs","p":pr},{"obs":x}); (y*d_rep).sum().backward()
eng = conv_encoder.ConvEncoderEngine(conv, torch.device(DEV))
out = torch.zeros(B, emb, device=DEV)
eng.forward(x.to(DEV), out, emb, save=True)
eng.backward(d_rep.to(DEV))
for k,key in enumerate(case_runner.ENC_KEYS[kind]):
    g = eng._seg(k, eng.grads).view(pr[key].shape).cpu(); ref = pr[key].grad
    d=(g-ref).double()
    print(key, "relL2 %.3e worst %.3e refnorm %.3e" % (float(d.norm()/(ref.double().norm()+1e-30)), float(d.abs().max()/(ref.abs().max()+1e-30)), float(ref.norm())))
# ---- intermediate: gradient wrt conv4 pre-activation
import torch.nn.functional as F
pr2 = {k: v.clone() for k,v in p.items()}
xx = x/255.0-0.5
a1 = F.relu(F.conv2d(xx, pr2["c1w"], pr2["c1b"], stride=2)); a2 = F.relu(F.conv2d(a1, pr2["c2w"], pr2["c2b"])); a3 = F.relu(F.conv2d(a2, pr2["c3w"], pr2["c3b"]))
z4 = F.conv2d(a3, pr2["c4w"], pr2["c4b"]).requires_grad_(True)
a4 = F.relu(z4); flat = a4.reshape(B,-1); zz = F.linear(flat, pr2["fcw"], pr2["fcb"]); zz.retain_grad()
yy = torch.tanh(F.layer_norm(zz, (emb,), pr2["lnw"], pr2["lnb"], 1e-5)); (yy*d_rep).sum().backward()
gz4 = z4.grad  # (B,32,35,35)
dy4 = [v for k_,v in eng.ws._bufs.items() if k_[0]=="b.dy1" and k_[1]==(B*35*35*32,)][0].view(B,35,35,32).permute(0,3,1,2).cpu()
d = (dy4-gz4).double(); print("dY4 relL2 %.3e worst %.3e" % (float(d.norm()/gz4.double().norm()), float(d.abs().max()/gz4.abs().max())))
dz_e = [v for k_,v in eng.ws._bufs.items() if k_[0]=="b.dz"][0].cpu(); d=(dz_e-zz.grad).double(); print("dz relL2 %.3e" % float(d.norm()/zz.grad.double().norm()))
dcolf = [v for k_,v in eng.ws._bufs.items() if k_[0]=="b.dcolf"][0].view(B,-1).cpu()
ref_dflat = zz.grad @ pr2["fcw"]
d=(dcolf-ref_dflat).double(); print("dcolf relL2 %.3e worst %.3e" % (float(d.norm()/ref_dflat.double().norm()), float(d.abs().max()/ref_dflat.abs().max())))
bad = (d.abs() > 1e-3*float(ref_dflat.abs().max())).nonzero()
print("bad count", len(bad), bad[:10].tolist())
y4e = eng.saved["ys"][-1].view(B,35,35,32).permute(0,3,1,2).cpu()
mm = ((y4e>0) != (z4.detach()>0))
print("mask mismatches", int(mm.sum()), "of", mm.numel(), "max |z4| at mismatches", float(z4.detach()[mm].abs().max()) if mm.any() else 0.0)
print("Y4 fwd max abs diff", float((y4e - a4.detach()).abs().max()))
d2 = (dy4-gz4)[~mm].double(); print("dY4 error excluding mismatched-mask elements: max", float(d2.abs().max()))
