"""Debug probe for the 32 x 64 `conv_wino_x3` tile's co-residency hazard: an AGGRESSOR convolution loops on one stream while a
VICTIM kernel loops on another; every victim launch is compared with the result it gives alone.  Prints, per (aggressor, victim) pair, how
many of the victim's launches differ.  Run on the GPU box: python tools/micro/coreside.py"""
import os
import sys
from ctypes import byref, c_void_p

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from airv2x_perception_amd import _lib                                                   # noqa: E402
from airv2x_perception_amd.opencood_iface.packing import pack_conv_weight                # noqa: E402

lib = _lib.load()
P = lambda t: c_void_p(t.data_ptr()) if t is not None else None
S = lambda s: c_void_p(s.cuda_stream)


def make_conv(n, h, w, cin, cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    wp, coutp = pack_conv_weight(wt)
    wp = wp.cuda()
    st = S(torch.cuda.current_stream())
    u = torch.empty(lib.av2x_wino_weight_bytes(cin, coutp) // 4, device="cuda")
    _lib.check(lib.av2x_wino_pack_weights(P(wp), cin, coutp, P(u), st), "pack")
    u3 = torch.empty(lib.av2x_wino_x3_weight_bytes(cin, coutp) // 2, dtype=torch.bfloat16, device="cuda")
    _lib.check(lib.av2x_wino_x3_pack_weights(P(wp), cin, coutp, P(u3), st), "pack3")
    scale, shift = torch.ones(cout).cuda(), torch.zeros(cout).cuda()
    torch.cuda.synchronize()
    return dict(x=x, wp=wp, u=u, u3=u3, scale=scale, shift=shift, n=n, h=h, w=w, cin=cin, cout=cout, coutp=coutp)


TILES = {"x3_32": 0x40000400 | (32 << 16) | 64, "x3_64": 0x40000400 | (64 << 16) | 64, "f32_h": 0x40000000 | (32 << 16) | 64 | 0x8000,
         "f32_q": 0x40000000 | (32 << 16) | 32 | 0x8000, "igemm": 0}


def run_conv(c, tile, out, stream):
    d = _lib.ConvDesc(n=c["n"], h=c["h"], w=c["w"], cin=c["cin"], in_ctot=c["cin"], in_coff=0, ho=c["h"], wo=c["w"], cout=c["cout"], coutp=c["coutp"],
                      out_ctot=c["cout"], out_coff=0, ks=3, stride=1, pad=1, relu=1, mode=0, up=1, tile=tile, sk_wgs=0)
    wgt = c["u3"] if tile & 0x400 else (c["u"] if tile & 0x40000000 else c["wp"])
    _lib.check(lib.av2x_conv2d_res(byref(d), P(c["x"]), P(wgt), P(c["scale"]), P(c["shift"]), None, P(out), S(stream)), "conv")


def make_linear(m_imgs, h, w, cin, cout, seed):
    from airv2x_perception_amd.opencood_iface.packing import to_bf16x3_koct
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(m_imgs, h, w, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 1, 1, generator=g) / np.sqrt(cin)
    wp, coutp = pack_conv_weight(wt)
    wp = wp.cuda()
    return dict(x=x, wp=wp, w3=to_bf16x3_koct(wp), scale=torch.ones(cout).cuda(), shift=torch.zeros(cout).cuda(), n=m_imgs, h=h, w=w, cin=cin, cout=cout, coutp=coutp)


def run_linear(c, tile, out, stream):
    d = _lib.ConvDesc(n=c["n"], h=c["h"], w=c["w"], cin=c["cin"], in_ctot=c["cin"], in_coff=0, ho=c["h"], wo=c["w"], cout=c["cout"], coutp=c["coutp"],
                      out_ctot=c["cout"], out_coff=0, ks=1, stride=1, pad=0, relu=0, mode=0, up=1, tile=tile, sk_wgs=0)
    wgt = c["w3"] if tile & 0x400 else c["wp"]
    _lib.check(lib.av2x_conv2d_res(byref(d), P(c["x"]), P(wgt), P(c["scale"]), P(c["shift"]), None, P(out), S(stream)), "linear")


def victims():
    g = torch.Generator().manual_seed(5)
    L, nv, H, W, heads = int(os.environ.get("FAX_L", "7")), int(os.environ.get("FAX_NV", "4")), 100, 352, 8
    qkv = torch.randn(L * H * W, 3 * heads * 32, generator=g).cuda()
    table = (torch.randn((2 * L - 1) * 49, heads, generator=g) * 0.1).cuda()
    out_a = torch.empty(L * H * W, heads * 32, device="cuda")

    def fax(stream, bits=0):
        _lib.check(lib.av2x_fax_attention(P(qkv), P(table), P(out_a), L, nv, H, W, 4, heads, 32, bits, S(stream)), "fax")
        return out_a
    xs = torch.randn(8 * 48 * 88, 256, generator=g).cuda()
    gam, bet = torch.randn(256, generator=g).cuda(), torch.randn(256, generator=g).cuda()
    out_l = torch.empty_like(xs)

    def ln(stream):
        _lib.check(lib.av2x_layernorm(P(xs), P(gam), P(bet), P(out_l), xs.shape[0], 256, 1e-5, S(stream)), "ln")
        return out_l
    cv = make_conv(2, 25, 88, 128, 128, 9)
    out_c = torch.empty(2, 25, 88, 128, device="cuda")

    def conv_f32(stream):
        run_conv(cv, TILES["f32_q"], out_c, stream)
        return out_c

    def conv_ig(stream):
        run_conv(cv, TILES["igemm"], out_c, stream)
        return out_c
    import ctypes
    gl = None
    gpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libablate_guard.so")
    if os.path.exists(gpath):
        gl = ctypes.CDLL(gpath)
    err = torch.zeros(4, dtype=torch.int32, device="cuda")

    def guard(stream):
        gl.guard_launch(c_void_p(err.data_ptr()), 512, 20480, 200, c_void_p(stream.cuda_stream))
        return err
    extra = {"guard_lds20k": guard} if gl is not None else {}
    if gl is not None:
        couts = {k: torch.zeros(1024 * 256, device="cuda") for k in range(15)}

        def chain(kind):
            def f(stream):
                gl.chain_launch(kind, c_void_p(couts[kind].data_ptr()), 1024, 64, 40, c_void_p(stream.cuda_stream))
                return couts[kind]
            return f
        extra.update({"chain_mfma16x16x4f32": chain(0), "chain_mfma32x32x2f32": chain(1), "chain_exp_shfl": chain(2), "chain_valu": chain(3),
                      "chain_pk_add_opsel": chain(4), "chain_pk_add_plain": chain(5), "chain_pkv_add_hi10": chain(6), "chain_pkv_add_sel01": chain(7),
                      "chain_pkv_mul_swap": chain(8), "chain_pkv_fma_hi101": chain(9), "chain_pkv_fma_sel010": chain(10),
                      "chain_pkv_add_neg": chain(11), "chain_pkv_add_swap_src0": chain(12), "chain_pkv_f16_sel01": chain(13),
                      "chain_pkv_u16_sel01": chain(14)})
    if os.environ.get("ONLY"):
        extra = {k: v for k, v in extra.items() if os.environ["ONLY"] in k}
    if os.environ.get("ONLY"):
        return {**extra, "fax_wave": lambda s: fax(s, 0)}
    return {**extra, "fax_wave": lambda s: fax(s, 0), "fax_mfma4": lambda s: fax(s, 8), "fax_generic": lambda s: fax(s, 4), "layernorm": ln,
            "conv_wino_f32_q": conv_f32, "conv_igemm_f32": conv_ig}


def main():
    reps = int(os.environ.get("REPS", "40"))
    agg = make_conv(4, 25, 88, 256, 256, 3)
    agg_out = torch.empty(4, 25, 88, 256, device="cuda")
    lin = make_linear(7, 100, 352, 256, 256, 4)
    lin_out = torch.empty(7, 100, 352, 256, device="cuda")
    X3P = {"x3p64": (128 << 16) | 64 | 0x1400, "x3p128": (128 << 16) | 128 | 0x1400, "ig_f32": 0}
    sa, sv = torch.cuda.Stream(), torch.cuda.Stream()
    MM = {}
    MICRO = torch.zeros(4096 * 256, device="cuda")
    for nm, dt in (("mm_bf16", torch.bfloat16), ("mm_f16", torch.float16), ("mm_f32", torch.float32)):
        a_, b_ = torch.randn(8192, 1024, device="cuda").to(dt), torch.randn(1024, 1024, device="cuda").to(dt)
        MM[nm] = (a_, b_, torch.empty(8192, 1024, device="cuda", dtype=dt))
    V = victims()
    run_conv(agg, TILES["x3_64"], agg_out, torch.cuda.current_stream())
    torch.cuda.synchronize()
    agg_ref = agg_out.clone()
    for vname, vfn in V.items():
        ref = vfn(torch.cuda.current_stream()).clone()
        torch.cuda.synchronize()
        for aname in os.environ.get("AGG", "x3p128,x3_32,f32_h,none").split(","):
            bad = 0
            worst = 0.0
            agg_bad = 0
            for _ in range(reps):
                def aggress():
                    if aname.startswith("micro_"):       # one instruction class per kernel (tools/micro/guard.hip aggressor_kernel)
                        import ctypes
                        gl_ = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libablate_guard.so"))
                        kind = {"micro_cvt_pk_bf16": 0, "micro_mfma_bf16": 1, "micro_mfma_f32": 2, "micro_cvt_vec": 3}[aname]
                        gl_.aggressor_launch(kind, c_void_p(MICRO.data_ptr()), 4096, 20000 if kind in (0, 3) else 3000, c_void_p(sa.cuda_stream))
                    elif aname.startswith("mm_"):          # a library GEMM (rocBLAS / hipBLASLt through torch) on the aggressor stream
                        with torch.cuda.stream(sa):
                            for _k in range(4):
                                torch.matmul(MM[aname][0], MM[aname][1], out=MM[aname][2])
                    elif aname in X3P:
                        for _k in range(2):
                            run_linear(lin, X3P[aname], lin_out, sa)
                    elif aname != "none":
                        for _k in range(6):
                            run_conv(agg, TILES[aname], agg_out, sa)
                aggress()
                out = vfn(sv)
                aggress()
                torch.cuda.synchronize()
                if not torch.equal(out, ref) and vname == "fax_wave" and os.environ.get("ANALYZE") and bad == 0:
                    L_, H_, W_ = int(os.environ.get("FAX_L", "7")), 100, 352
                    d = (out != ref).view(L_, H_, W_, 8, 32)
                    print("   wrong elements", int(d.sum()), "of", d.numel(), "nan", int(torch.isnan(out).sum()))
                    print("   by query agent:", d.sum(dim=(1, 2, 3, 4)).tolist())
                    print("   by head:", d.sum(dim=(0, 1, 2, 4)).tolist())
                    print("   by dim:", d.sum(dim=(0, 1, 2, 3)).tolist())
                    rows = d.sum(dim=(0, 2, 3, 4))
                    print("   by pixel row (first 40):", rows[:40].tolist())
                    cols = d.sum(dim=(0, 1, 3, 4))
                    print("   by pixel col (first 48):", cols[:48].tolist())
                    wx = d.view(L_, 25, 4, 88, 4, 8, 32).sum(dim=(0, 2, 4, 5, 6))
                    print("   windows touched:", int((wx > 0).sum()), "of", wx.numel(), "; first touched window ids:", (wx.reshape(-1) > 0).nonzero().reshape(-1)[:40].tolist())
                    e = (out - ref).abs()
                    print("   error magnitude: max", float(e.max()), "median of wrong", float(e[out != ref].median()))
                if vname == "fax_wave" and hasattr(lib, "av2x_dbg_fax"):
                    import ctypes, struct
                    buf = (ctypes.c_uint * (8 + 320))()
                    lib.av2x_dbg_fax(buf, 0)
                    if buf[0] and bad < 2:
                        print("   in-kernel DS-vs-global bias mismatches:", buf[0])
                        for sl in range(min(int(buf[0]), 12)):
                            d = buf[8 + sl * 8: 16 + sl * 8]
                            f = lambda u: struct.unpack("f", struct.pack("I", u))[0]
                            print(f"     wg {d[0]} thread {d[1]} (wave {d[1] >> 6} lane {d[1] & 63}) idx {d[2]} head {d[5]} qt {d[6]}: ds {f(d[3]):+.6f} global {f(d[4]):+.6f} ds re-read {f(d[7]):+.6f}")
                    lib.av2x_dbg_fax(buf, 1)
                if not torch.equal(out, ref):
                    bad += 1
                    worst = max(worst, float((out - ref).abs().max()))
                    if vname.startswith("guard"):
                        ref = out.clone()
                if aname in ("x3_32",) and not torch.equal(agg_out, agg_ref):
                    agg_bad += 1
            if vname.startswith("guard"):
                print("   guard error counters [lds words, vgprs, agprs, -]:", out.tolist(), flush=True)
                out.zero_()
            print(f"victim {vname:16s} aggressor {aname:6s}: {bad:3d} / {reps} victim launches differ (max abs {worst:.3e}); aggressor's own output differs {agg_bad}",
                  flush=True)


if __name__ == "__main__":
    main()
