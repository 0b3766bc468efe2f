"""SURVEY.md section 8(d) sweep: the ensemble-Q kernel (forward of N critics on one batch) and the fused
critic forward+backward kernel over batch sizes and ensemble sizes, each against both rooflines.

    python tools/sweep.py  > profiles/rN_sweep.md      (on the GPU box)

Algorithmic bytes / flops follow SURVEY.md 8(d):
    ensemble-Q : bytes = N*(in*H + H^2 + 3H + 1)*4 + B*in*4 + N*B*4 ; flops = 2*B*N*(in*H + H^2 + H)
    critic f+b : flops = 2*B*N*(in*H + 2*H^2 + H)   (fc1, fc2, head, fc2 backward-data)
                 bytes = ensemble-Q bytes + stores of h1, h2, dz2, dz1 (4*N*B*H*4) + B*4 (td)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import super_sac_amd as ssa  # noqa: E402

PEAK_TF, PEAK_GBS = 157.3, 8000.0
dev = torch.device("cuda")
lib = ssa._lib.lib


def timed(fn, reps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def arena_for(N, in_dim, H, out):
    ar = ssa.engine.MlpArena(N, in_dim, H, out, dev)
    g = torch.Generator(device="cpu").manual_seed(1)
    for j in range(N):
        for seg in ssa.engine.SEGS:
            v = ar.view(j, seg)
            v.copy_(torch.randn(v.shape, generator=g) * 0.05)
    return ar


def main():
    tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0  # 0 auto, 16, 17 (16 rows single-buffered), 32
    ssa._lib.check(lib.ssac_fused_tile_rows(tile))
    print(f"tile variant {tile} (0 = automatic)\n")
    print("| kernel | obs+act | H | N | B | us/launch | TFLOP/s | % fp32 MFMA peak | alg. GB/s | % HBM peak |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    ws = ssa.engine.Workspace(dev)
    for (in_dim, H) in [(23, 256), (393, 256)][: (1 if tile else 2)]:
        for N in ((10,) if tile else (10, 16)):
            ar = arena_for(N, in_dim, H, 1)
            for B in (256, 512, 4096, 16384, 65536):
                if in_dim == 393 and B > 16384:
                    continue
                x = torch.randn(B, in_dim, device=dev)
                reps = 200 if B <= 4096 else 30
                t = timed(lambda: ssa.engine.mlp_forward(ar, x, in_dim, 0, B, ws, f"q{B}", save=False), reps)
                fl = 2.0 * B * N * (in_dim * H + H * H + H)
                by = N * (in_dim * H + H * H + 3 * H + 1) * 4 + B * in_dim * 4 + N * B * 4
                print(f"| ensemble-Q fwd | {in_dim} | {H} | {N} | {B} | {t*1e6:.1f} | {fl/t/1e12:.1f} | "
                      f"{100*fl/t/1e12/PEAK_TF:.1f} | {by/t/1e9:.0f} | {100*by/t/1e9/PEAK_GBS:.1f} |")
                # fused forward+backward of the critic loss
                td = torch.randn(B, 1, device=dev)
                h1 = torch.empty(N, B, H, device=dev)
                h2 = torch.empty_like(h1)
                dz2 = torch.empty_like(h1)
                dz1 = torch.empty_like(h1)
                q = torch.empty(N, B, 1, device=dev)
                dq = torch.empty_like(q)
                tiles = int(lib.ssac_fused_row_tiles(C.byref(ar.desc()), B, N))
                parts = torch.zeros(N * tiles * 2, device=dev)

                def run():
                    ssa._lib.check(lib.ssac_critic_fwd_bwd_fused(
                        C.byref(ar.desc()), x.data_ptr(), in_dim, B, td.data_ptr(), 0, 0, 1, 0, 0, float(N),
                        h1.data_ptr(), h2.data_ptr(), q.data_ptr(), dq.data_ptr(), dz2.data_ptr(),
                        dz1.data_ptr(), parts.data_ptr(), 0, ssa.engine.stream()))
                t = timed(run, reps)
                fl = 2.0 * B * N * (in_dim * H + 2 * H * H + H)
                by = by + 4 * N * B * H * 4 + B * 4
                print(f"| critic fwd+bwd | {in_dim} | {H} | {N} | {B} | {t*1e6:.1f} | {fl/t/1e12:.1f} | "
                      f"{100*fl/t/1e12/PEAK_TF:.1f} | {by/t/1e9:.0f} | {100*by/t/1e9/PEAK_GBS:.1f} |")
                del h1, h2, dz2, dz1
            del ar
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
