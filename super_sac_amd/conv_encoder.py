"""Pixel encoders on the HIP path (reference super_sac/nets/cnns.py:37-103).

``BigPixelEncoder`` (DrQ: conv3x3 s2 + 3x conv3x3 s1, ReLU, fc, LayerNorm, tanh; input x/255-0.5) and
``SmallPixelEncoder`` (Nature-DQN: conv8 s4, conv4 s2, conv3 s1, ReLU, fc; input x/255) run as

    im2col (normalisation fused) -> exact-fp32 MFMA GEMM with bias+ReLU epilogue      per conv layer
    im2col with kernel = whole map -> GEMM                                            fc over the NCHW flatten
    LayerNorm+tanh kernel                                                             (Big only)

with channels-last activations, so each GEMM output is the next layer's input and the
``nn.Conv2d`` weights are used in place as (out, c*kh*kw) matrices.  The backward pass mirrors it
(GEMM backward-data -> col2im with the ReLU mask fused; split-K weight-gradient GEMMs reduced in a
fixed order), leaving the gradients in one flat arena so clip_grad_norm_ and Adam are two launches.
The module's parameters are re-pointed at a flat arena exactly like the MLP ensembles.
"""
import ctypes as C

import torch

from . import engine
from ._lib import check, lib

USE_IMPLICIT = True          # implicit-GEMM inner conv layers (False: every layer as im2col + GEMM; the tests' A/B)
USE_IMPLICIT_FIRST = True    # the first layer as an implicit GEMM over the NCHW image
FC_CHANNELS_LAST = True      # fc reads the last map in place
IMPLICIT_MIN_ROWS = 16_384       # output pixels (B*Ho*Wo) from which the implicit-GEMM kernels pay off
IMPLICIT_ROWS_PER_SLICE = 256    # output pixels per weight-gradient slice (= per workgroup), at least
FIRST_MIN_ROWS = 16_384          # output pixels from which the first layer leaves im2col
FIRST_ROWS_PER_SLICE = 512       # output pixels per first-layer weight-gradient slice, at least
IMPLICIT_WG_PER_CU = 2           # weight-gradient workgroups per CU (248 VGPRs: at most two waves per SIMD) ...
IMPLICIT_WG_BIG_ROWS = 300_000   # ... and ONE from this many output pixels on (measured: DMC 3.31 -> 3.25 ms; fewer,
                                 # longer slices amortise the cross-wave sum and the partial-slice traffic)
FIRST_WG_PER_CU = 2              # (measured 2 / 3 / 4 / 6: Atari 1.57 / 1.59 / 1.61 / 1.61 ms)
FIRST_WGRAD_BANDS = True         # conv1's weight gradient with the image staged through LDS where covered (DMC: 116 -> 86 us)
WGRAD_WHOLE_IMAGES = True        # weight gradients of layers with small feature maps: x and dy of an image staged in LDS
FC_STREAM = True  # the fc forward as an operand stream (N <= 64 outputs; DMC 49 -> 38 us per pass)
FC_SLICES = 48  # K slices of the tiled fc forward (8 row tiles x 48 slices ~ 1.5 workgroups per CU at B 512)
ROWS_PER_SLICE = 4096  # split-K granularity of the convolution weight gradients


def find_conv_module(encoder):
    """the BigPixelEncoder/SmallPixelEncoder-shaped module inside an Encoder wrapper
    (train_dmc_from_pixels.py:15-27 ``conv_block``, train_atari.py:9-20 ``cnn``)."""
    for m in encoder.modules():
        if hasattr(m, "conv1") and hasattr(m, "fc"):
            return m
    return None


class ConvEncoderEngine:
    def __init__(self, module, device):
        self.module = module
        self.device = device
        convs = [getattr(module, n) for n in ("conv1", "conv2", "conv3", "conv4") if hasattr(module, n)]
        self.convs = convs
        self.big = hasattr(module, "ln")
        self.geom = [(c.in_channels, c.out_channels, c.kernel_size[0], c.stride[0]) for c in convs]
        self.emb = module.fc.out_features
        # inner layers with 32-multiple channel counts run as implicit GEMMs (csrc/ssac_conv_implicit.hip)
        # (small maps -- decided per call from the row count -- stay on im2col; a strided layer's backward-data pass
        # runs per parity class of input pixels, so no MFMA is spent on a structurally zero tap)
        self.implicit_ok = [USE_IMPLICIT and l > 0 and bool(lib.ssac_conv_implicit_supported(ci, co, k))
                            for l, (ci, co, k, s) in enumerate(self.geom)]
        self.implicit = list(self.implicit_ok)
        self.first = False   # the saved forward ran the first layer as an implicit GEMM
        self.div, self.shift = (255.0, -0.5) if self.big else (255.0, 0.0)
        # ---- flat parameter arena (each tensor starts at a multiple of 4 floats)
        plist = []
        for c in convs:
            plist += [c.weight, c.bias]
        plist += [module.fc.weight, module.fc.bias]
        if self.big:
            plist += [module.ln.weight, module.ln.bias]
        self.plist = plist
        offs, o = [], 0
        for p in plist:
            offs.append(o)
            o += (p.numel() + 3) // 4 * 4
        self.offs, self.numel = offs, o
        self.flat = torch.zeros(o, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p, off in zip(plist, offs):
                v = self.flat[off:off + p.numel()].view(p.shape)
                v.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = v
        self.grads = torch.zeros_like(self.flat)
        self.ws = engine.Workspace(device)
        self.saved = None

    def is_bound(self):
        return all(p.data_ptr() == self.flat.data_ptr() + 4 * off for p, off in zip(self.plist, self.offs))

    def _seg(self, k, tensor=None):
        t = self.flat if tensor is None else tensor
        return t[self.offs[k]:self.offs[k] + self.plist[k].numel()]

    # ------------------------------------------------------------------------------------
    def forward(self, img, dst, ld_dst, save):
        """img (B, C, H, W) fp32 on the device; writes the embedding into dst[:, :emb] (row stride
        ld_dst).  With `save`, keeps what backward() needs."""
        engine.require_gpu(img)
        img = img.contiguous()
        self.forward_ptr(img.data_ptr(), tuple(img.shape), 1 if img.dtype == torch.uint8 else 0, dst.data_ptr(), ld_dst, save,
                         img=img, dst=dst)

    def forward_ptr(self, img_ptr, shape, u8, dst_ptr, ld_dst, save, img=None, dst=None):
        """forward() on a raw device pointer: (B, C, H, W) fp32 -- or uint8 (u8 = 1: the cast and the input normalisation
        happen in the first layer's patch gather) -- e.g. the observation buffer of an acting plan (acting.py), which is
        not a torch allocation.  Every launch reads / writes workspace buffers keyed by shape: recordable."""
        B, Cc, Hh, Ww = shape
        st = engine.stream()
        src_ptr = img_ptr
        strides = (Cc * Hh * Ww, Hh * Ww, Ww, 1)
        Hi, Wi, div, shift = Hh, Ww, self.div, self.shift
        cols, ys, shapes = [], [], []
        tag = "s" if save else "t"
        for l, (ci, co, k, s) in enumerate(self.geom):
            Ho, Wo = (Hi - k) // s + 1, (Wi - k) // s + 1
            rows, ckk = B * Ho * Wo, ci * k * k
            y = self.ws.get(f"{tag}.y{l if save else l % 2}", (rows * co,))
            if save:
                self.implicit[l] = self.implicit_ok[l] and rows >= IMPLICIT_MIN_ROWS
            first = (l == 0 and USE_IMPLICIT_FIRST and rows >= FIRST_MIN_ROWS and not u8 and img_ptr % 16 == 0
                     and lib.ssac_conv_first_supported(ci, co, k, s, Hi, Wi, B) > 0)
            if l == 0 and save:
                self.first = first
            if first:
                # the gather AND the input normalisation happen in the operand loads; the image is what backward reads
                col = None
                check(lib.ssac_conv_first_fwd(img_ptr, self.convs[0].weight.data_ptr(),
                                              self.convs[0].bias.data_ptr(), y.data_ptr(), B, ci, Hi, Wi, co, k, s,
                                              div, shift, st))
            elif self.implicit_ok[l] and rows >= IMPLICIT_MIN_ROWS:
                # channels-last input straight from the previous layer: the patch gather happens in the operand
                # loads of the implicit-GEMM kernel, no column matrix
                col = None
                check(lib.ssac_conv_fwd(src_ptr, self.convs[l].weight.data_ptr(),
                                        self.convs[l].bias.data_ptr(), y.data_ptr(), B, Hi, Wi, ci, co, k, s, st))
            else:
                col = self.ws.get(f"{tag}.col{l if save else 0}", (rows * ckk,))
                check(lib.ssac_im2col(src_ptr, u8 if l == 0 else 0, *strides, B, ci, Hi, Wi, k, s, div, shift,
                                      col.data_ptr(), st))
                check(lib.ssac_linear_fwd(col.data_ptr(), ckk, self.convs[l].weight.data_ptr(), ckk,
                                          self.convs[l].bias.data_ptr(), y.data_ptr(), co, rows, co, ckk, 1, st))
            cols.append(col); ys.append(y); shapes.append((ci, co, k, s, Hi, Wi, Ho, Wo))
            src, src_ptr, strides = y, y.data_ptr(), (Ho * Wo * co, 1, Wo * co, co)  # channels-last view of the GEMM output
            Hi, Wi, div, shift = Ho, Wo, 1.0, 0.0
        co = self.geom[-1][1]
        flat_dim = co * Hi * Wi
        fc = self.module.fc
        if FC_CHANNELS_LAST:
            # the last feature map IS the fc input once the weight's columns are put in channels-last order
            # (emb x C x P -> emb x P x C, 7.8 MB for DrQ: cheaper than gathering B x C*P activations both ways)
            colf = src
            wfc = self.ws.get("fc.wcl", (self.emb * flat_dim,))
            check(lib.ssac_permute_cp(fc.weight.data_ptr(), wfc.data_ptr(), self.emb, co, Hi * Wi, 1, st))
        else:
            colf = self.ws.get(f"{tag}.colf", (B * flat_dim,))
            check(lib.ssac_im2col(src_ptr, 0, *strides, B, co, Hi, Wi, Hi, 1, 1.0, 0.0, colf.data_ptr(), st))
            wfc = fc.weight
        fc = self.module.fc

        def fc_forward(out_ptr, ld_out):
            # (B x flat_dim) . (emb x flat_dim)^T: 8 x 1 output tiles only -> cut K into slices so the launch fills
            # the chip, then a fixed-order reduction adds the bias
            # few outputs (N <= 64): the operand-stream kernel, one wave per (32 rows, K slice), one workgroup per CU
            row_groups = (B + 127) // 128
            skps = ((flat_dim + max(1, 256 // row_groups) - 1) // max(1, 256 // row_groups) + 7) // 8 * 8
            if FC_STREAM and lib.ssac_linear_fwd_stream_supported(B, self.emb, flat_dim, skps, flat_dim, flat_dim) \
                    and (flat_dim + skps - 1) // skps >= 4:
                slices = (flat_dim + skps - 1) // skps
                part = self.ws.get("fc.partial", (slices * B * self.emb,))
                check(lib.ssac_linear_fwd_stream(colf.data_ptr(), flat_dim, wfc.data_ptr(), flat_dim, part.data_ptr(), B,
                                                 self.emb, flat_dim, skps, st))
                check(lib.ssac_reduce_slices_bias(part.data_ptr(), slices, B, self.emb, fc.bias.data_ptr(), out_ptr,
                                                  ld_out, st))
                return
            kps = max(32, ((flat_dim + FC_SLICES - 1) // FC_SLICES + 31) // 32 * 32)
            slices = (flat_dim + kps - 1) // kps
            if slices < 4:
                check(lib.ssac_linear_fwd(colf.data_ptr(), flat_dim, wfc.data_ptr(), flat_dim,
                                          fc.bias.data_ptr(), out_ptr, ld_out, B, self.emb, flat_dim, 0, st))
                return
            part = self.ws.get("fc.partial", (slices * B * self.emb,))
            check(lib.ssac_linear_fwd_splitk(colf.data_ptr(), flat_dim, wfc.data_ptr(), flat_dim,
                                             part.data_ptr(), B, self.emb, flat_dim, kps, st))
            check(lib.ssac_reduce_slices_bias(part.data_ptr(), slices, B, self.emb, fc.bias.data_ptr(), out_ptr,
                                              ld_out, st))
        if self.big:
            z = self.ws.get(f"{tag}.z", (B, self.emb))
            fc_forward(z.data_ptr(), self.emb)
            xhat = self.ws.get(f"{tag}.xhat", (B, self.emb))
            rstd = self.ws.get(f"{tag}.rstd", (B,))
            ln = self.module.ln
            check(lib.ssac_ln_tanh_fwd(z.data_ptr(), self.emb, ln.weight.data_ptr(), ln.bias.data_ptr(), B,
                                       self.emb, dst_ptr, ld_dst, xhat.data_ptr(), rstd.data_ptr(), st))
        else:
            xhat = rstd = None
            fc_forward(dst_ptr, ld_dst)
        if save:
            self.saved = dict(B=B, img=img, cols=cols, ys=ys, shapes=shapes, colf=colf, wfc=wfc, flat_dim=flat_dim,
                              Hf=Hi, Wf=Wi, xhat=xhat, rstd=rstd, out=dst, ld_out=ld_dst)

    # ------------------------------------------------------------------------------------
    def backward(self, d_rep, accumulate=False):
        """d_rep (B, emb) contiguous = dL/d(embedding).  Fills self.grads (flat, same layout as the
        parameters); accumulate=True ADDS to it instead (the members of an ensemble share one encoder and the
        reference runs ONE backward over the sum of their losses, learning.py:47-121)."""
        if accumulate:
            keep, tmp = self.grads, self.__dict__.get("_gtmp")
            if tmp is None:
                tmp = self.__dict__["_gtmp"] = torch.zeros_like(self.flat)
            self.grads = tmp
            try:
                self.backward(d_rep)
            finally:
                self.grads = keep
            keep.add_(tmp)   # (device plumbing: one elementwise pass over the gradient arena)
            return
        sv = self.saved
        assert sv is not None, "backward() needs a forward(save=True)"
        B, st = sv["B"], engine.stream()
        nconv = len(self.geom)
        k_fcw, k_fcb = 2 * nconv, 2 * nconv + 1
        if self.big:
            dz = self.ws.get("b.dz", (B, self.emb))
            scratch = self.ws.get("b.lnscratch", (B, self.emb))
            check(lib.ssac_ln_tanh_bwd(d_rep.data_ptr(), self.emb, sv["out"].data_ptr(), sv["ld_out"],
                                       sv["xhat"].data_ptr(), sv["rstd"].data_ptr(),
                                       self.module.ln.weight.data_ptr(), B, self.emb, dz.data_ptr(), self.emb,
                                       scratch.data_ptr(), self._seg(k_fcb + 1, self.grads).data_ptr(),
                                       self._seg(k_fcb + 2, self.grads).data_ptr(), st))
        else:
            dz = d_rep
        flat_dim = sv["flat_dim"]
        ci, co, k, s, Hi, Wi, Ho, Wo = sv["shapes"][-1]
        dy = self.ws.get(f"b.dy{(nconv - 1) % 2}", (B * Ho * Wo * co,))
        ylast = sv["ys"][-1]
        if FC_CHANNELS_LAST:
            # fc weight gradient in channels-last column order, then back to the parameter's (C, P) order
            gcl = self.ws.get("fc.gcl", (self.emb * flat_dim,))
            check(lib.ssac_linear_wgrad_splitk(dz.data_ptr(), self.emb, sv["colf"].data_ptr(), flat_dim,
                                               gcl.data_ptr(), self._seg(k_fcb, self.grads).data_ptr(), self.emb,
                                               flat_dim, B, B, st))
            check(lib.ssac_permute_cp(gcl.data_ptr(), self._seg(k_fcw, self.grads).data_ptr(), self.emb, co,
                                      Ho * Wo, 0, st))
            # (the last map's ReLU derivative rides in the GEMM's epilogue: dy is written once, masked)
            check(lib.ssac_linear_dgrad_masked(dz.data_ptr(), self.emb, sv["wfc"].data_ptr(), flat_dim,
                                               ylast.data_ptr(), flat_dim, dy.data_ptr(), flat_dim, B, flat_dim,
                                               self.emb, st))
        else:
            # fc: weight gradient (K = B rows, one slice writes straight into the gradient arena)
            check(lib.ssac_linear_wgrad_splitk(dz.data_ptr(), self.emb, sv["colf"].data_ptr(), flat_dim,
                                               self._seg(k_fcw, self.grads).data_ptr(),
                                               self._seg(k_fcb, self.grads).data_ptr(), self.emb, flat_dim, B, B, st))
            dcolf = self.ws.get("b.dcolf", (B * flat_dim,))
            check(lib.ssac_linear_dgrad(dz.data_ptr(), self.emb, self.module.fc.weight.data_ptr(), flat_dim,
                                        dcolf.data_ptr(), flat_dim, B, flat_dim, self.emb, st))
            cl = (Ho * Wo * co, 1, Wo * co, co)
            check(lib.ssac_col2im(dcolf.data_ptr(), dy.data_ptr(), *cl, ylast.data_ptr(), *cl, B, co, Ho, Wo,
                                  sv["Hf"], 1, st))
        for l in range(nconv - 1, -1, -1):
            ci, co, k, s, Hi, Wi, Ho, Wo = sv["shapes"][l]
            rows, ckk = B * Ho * Wo, ci * k * k
            first = l == 0 and self.first
            if first or self.implicit[l]:
                # one slice = one workgroup per (ci/32, co/32) block pair: as many slices as are resident at once (a
                # partly filled second round costs a whole one), never below the configured slice size
                blocks = 1 if first else (ci // 32) * (co // 32)
                per_cu = FIRST_WG_PER_CU if first else 1 if rows >= IMPLICIT_WG_BIG_ROWS else IMPLICIT_WG_PER_CU
                resident = max(1, 256 * per_cu // blocks)
                rps = max(FIRST_ROWS_PER_SLICE if first else IMPLICIT_ROWS_PER_SLICE,
                          ((rows + resident - 1) // resident + 127) // 128 * 128)
            else:
                rps = ROWS_PER_SLICE
            slices = (rows + rps - 1) // rps
            band_slices = int(lib.ssac_conv_first_wgrad_band_slices(B, ci, Hi, Wi, co, k, s)) if first and FIRST_WGRAD_BANDS else 0
            if band_slices:   # (the image staged through LDS in bands of output rows: one partial per persistent workgroup)
                slices = band_slices
            img_slices = (int(lib.ssac_conv_wgrad_img_slices(B, Hi, Wi, ci, co, k, s))
                          if (not first and self.implicit[l] and WGRAD_WHOLE_IMAGES) else 0)
            if img_slices:    # (small feature maps: both operands of an image staged in LDS, one partial per workgroup)
                slices = img_slices
            pw = self.ws.get("b.pw", (slices * co * ckk,))
            pb = self.ws.get("b.pb", (slices * co,))
            if band_slices:
                check(lib.ssac_conv_first_wgrad_band(dy.data_ptr(), sv["img"].data_ptr(), pw.data_ptr(), pb.data_ptr(), B, ci,
                                                     Hi, Wi, co, k, s, self.div, self.shift, st))
            elif first:
                check(lib.ssac_conv_first_wgrad(dy.data_ptr(), sv["img"].data_ptr(), pw.data_ptr(), pb.data_ptr(), B, ci,
                                                Hi, Wi, co, k, s, self.div, self.shift, rps, st))
            elif img_slices:
                check(lib.ssac_conv_wgrad_img(dy.data_ptr(), sv["ys"][l - 1].data_ptr(), pw.data_ptr(), pb.data_ptr(), B, Hi,
                                              Wi, ci, co, k, s, st))
            elif self.implicit[l]:
                x_in = sv["ys"][l - 1]  # this layer's input = previous layer's ReLU output, channels-last
                check(lib.ssac_conv_wgrad(dy.data_ptr(), x_in.data_ptr(), pw.data_ptr(), pb.data_ptr(), B, Hi, Wi,
                                          ci, co, k, s, rps, st))
            else:
                check(lib.ssac_linear_wgrad_splitk(dy.data_ptr(), co, sv["cols"][l].data_ptr(), ckk, pw.data_ptr(),
                                                   pb.data_ptr(), co, ckk, rows, ROWS_PER_SLICE, st))
            check(lib.ssac_reduce_slices_pair(pw.data_ptr(), co * ckk, self._seg(2 * l, self.grads).data_ptr(),
                                              pb.data_ptr(), co, self._seg(2 * l + 1, self.grads).data_ptr(), slices, st))
            if l == 0:
                break
            pci, pco, pk, ps, pHi, pWi, pHo, pWo = sv["shapes"][l - 1]
            dprev = self.ws.get(f"b.dy{(l - 1) % 2}", (B * pHo * pWo * pco,))
            if self.implicit[l] and s <= 4:   # (strided layers: one parity class of input pixels per tile)
                check(lib.ssac_conv_dgrad(dy.data_ptr(), self.convs[l].weight.data_ptr(), sv["ys"][l - 1].data_ptr(),
                                          dprev.data_ptr(), B, Hi, Wi, ci, co, k, s, st))
            else:
                dcol = self.ws.get("b.dcol", (rows * ckk,))
                if ci % 4 == 0:
                    # columns in (ky, kx, c) order: the adjoint gather then reads 16 contiguous bytes per tap
                    wcl = self.ws.get("b.wcl", (co * ckk,))
                    check(lib.ssac_permute_cp(self.convs[l].weight.data_ptr(), wcl.data_ptr(), co, ci, k * k, 1, st))
                    check(lib.ssac_linear_dgrad(dy.data_ptr(), co, wcl.data_ptr(), ckk, dcol.data_ptr(), ckk, rows,
                                                ckk, co, st))
                    check(lib.ssac_col2im_cl(dcol.data_ptr(), dprev.data_ptr(), sv["ys"][l - 1].data_ptr(), B, ci, Hi,
                                             Wi, k, s, st))
                else:
                    check(lib.ssac_linear_dgrad(dy.data_ptr(), co, self.convs[l].weight.data_ptr(), ckk,
                                                dcol.data_ptr(), ckk, rows, ckk, co, st))
                    pcl = (pHo * pWo * pco, 1, pWo * pco, pco)
                    check(lib.ssac_col2im(dcol.data_ptr(), dprev.data_ptr(), *pcl, sv["ys"][l - 1].data_ptr(), *pcl,
                                          B, ci, Hi, Wi, k, s, st))
            dy = dprev

    # ------------------------------------------------------------------------------------
    def optimizer_step(self, optimizer, clip, norm_out=None, step=True):
        """clip_grad_norm_(encoder.parameters(), clip) + encoder_optimizer.step() (learning.py:127-129).
        step=False: clip and log only (offline_actor_update without update_encoder, learning.py:203-208)."""
        st = engine.stream()
        adam = engine.adam_group(optimizer, self.device)
        if step:
            adam.advance()
        nb = int(lib.ssac_sumsq_blocks())
        ss = self.ws.get("o.ss", (nb,))
        check(lib.ssac_sumsq(self.grads.data_ptr(), self.numel, ss.data_ptr(), st))
        check(lib.ssac_clip_coef(adam.ctl.ptr, ss.data_ptr(), nb, float(clip) if clip else 0.0, 0, st))
        if norm_out is not None:
            check(lib.ssac_group_norms(ss.data_ptr(), 1, nb, adam.ctl.ptr, norm_out.data_ptr(), st))
        if not step:
            return
        m, v = adam.moments_for("conv_encoder", self.flat)
        check(lib.ssac_adam_step(self.flat.data_ptr(), m.data_ptr(), v.data_ptr(), self.grads.data_ptr(),
                                 self.numel, adam.ctl.ptr, st))


def conv_engine(encoder, device):
    """ConvEncoderEngine of `encoder` (cached on the wrapped conv module; re-packed after a deepcopy)."""
    mod = find_conv_module(encoder)
    if mod is None:
        return None
    eng = mod.__dict__.get("_ssac_conv")
    if eng is None or eng.module is not mod or not eng.is_bound() or eng.device != device:
        eng = ConvEncoderEngine(mod, device)
        mod.__dict__["_ssac_conv"] = eng
    return eng
