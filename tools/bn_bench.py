"""The HBM-bound training kernels on the BEV backbone's map shapes: microseconds and achieved GB/s of the algorithmic bytes
(stats: z read once; normalise: z read + y written; backward: dy and z read twice, dz written; pixel-attention backward:
n maps + dout read, n gradient maps written)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from airv2x_perception_amd.opencood_iface import train_ops as T


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    from airv2x_perception_amd import _lib
    from airv2x_perception_amd.opencood_iface.autograd import _P, _runner
    dev = torch.device("cuda", 0)
    r = _runner(dev)
    lib, st = r.lib, r.stream()
    for n, h, w, c in [(4, 100, 352, 64), (4, 50, 176, 128), (4, 25, 88, 256), (1, 100, 352, 128)]:
        z = torch.randn(n, h, w, c, device=dev)
        dy = torch.randn(n, h, w, c, device=dev)
        gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
        mb = z.numel() * 4 / 1e6
        rows = z.numel() // c
        mean, var, cnt = T.bn_stats(z)
        rstd, scale, shift = T.bn_finalize(mean, var, cnt, gamma, beta, 1e-3)
        ws = torch.empty(int(lib.av2x_bn_workspace_bytes(rows, c)) // 8 + 1, dtype=torch.float64, device=dev)
        y, dz, dg, db = torch.empty_like(z), torch.empty_like(z), torch.empty(c, device=dev), torch.empty(c, device=dev)
        # pre-allocated buffers, raw C-ABI calls: the launches queue back to back, the GPU (not the host) sets the pace
        t1 = timed(lambda: lib.av2x_bn_stats(_P(z), rows, c, _P(ws), _P(mean), _P(var), st))
        t2 = timed(lambda: lib.av2x_affine_act(_P(z), rows, c, _P(scale), _P(shift), 1, _P(y), st))
        t3 = timed(lambda: lib.av2x_bn_backward(_P(dy), _P(z), rows, c, _P(mean), _P(rstd), _P(scale), _P(shift), 1, _P(ws), _P(dg), _P(db), _P(dz), st))
        print(f"{n}x{h}x{w}x{c} ({mb:.0f} MB): stats {t1:6.1f} us {mb / t1 * 1e3:6.0f} GB/s | normalise+ReLU {t2:6.1f} us {2 * mb / t2 * 1e3:6.0f} GB/s | "
              f"backward {t3:6.1f} us {5 * mb / t3 * 1e3:6.0f} GB/s")
    from ctypes import c_void_p
    for k, h, w, c in [(4, 100, 352, 64), (4, 50, 176, 128), (4, 25, 88, 256)]:
        x = torch.randn(k, h, w, c, device=dev)
        g = torch.randn(h, w, c, device=dev)
        dx = torch.empty_like(x)
        arr = (c_void_p * k)(*[x[j].data_ptr() for j in range(k)])
        darr = (c_void_p * k)(*[dx[j].data_ptr() for j in range(k)])
        mb = x.numel() * 4 / 1e6
        t = timed(lambda: lib.av2x_pixel_attn_backward(arr, k, h * w, c, _P(g), darr, st))
        print(f"pixel attention backward {k} agents {h}x{w}x{c}: {t:6.1f} us {(2 * mb + mb / k) / t * 1e3:6.0f} GB/s")


if __name__ == "__main__":
    main()
