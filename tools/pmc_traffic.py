#!/usr/bin/env python3
"""Turn rocprofv3 PMC passes into per-launch HBM-side traffic for every conv_igemm_f32 variant.

  python tools/pmc_traffic.py --fetch gpurun_out/pmc3_bench_FETCH_SIZE --write gpurun_out/pmc3_bench_WRITE_SIZE \\
         --calib-fetch gpurun_out/pmc3_calib_FETCH_SIZE --calib-write gpurun_out/pmc3_calib_WRITE_SIZE \\
         -o profiles/pmc_hbm.json

Each directory holds the sqlite output of ONE `rocprofv3 --kernel-trace --pmc <counter>` run (separate
passes: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2).  Units and correction follow
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are KiB; on gfx950 FETCH_SIZE reports
half of the bytes of 16 B/lane streaming reads, WRITE_SIZE is "uncalibrated".  tools/pmc_calib.py launches
kernels with known byte counts; the scale factors measured there are stored in the output and applied."""
import argparse, glob, json, re, sqlite3
from collections import defaultdict


def rows(d):
    f = glob.glob(d.rstrip("/") + "/*/*_results.db") + glob.glob(d.rstrip("/") + "/*_results.db")
    c = sqlite3.connect(f[0])
    return c.execute("select kernel_name, grid_size, workgroup_size, counter_name, value from counters_collection "
                     "order by dispatch_id").fetchall()


def conv_key(name):
    g = re.search(r"conv_igemm_f32_glds<(\d+), (\d+), (\d+), (\d+), (\d), (\d)>", name)
    if g:   # LDS-DMA kernels: bench.py's key "g<BM>x<BN>[w8][d = 3 LDS stages][sk]"
        bm, bn, wm, wn, stages, mode = (int(g.group(i)) for i in range(1, 7))
        waves = (bm // wm) * (bn // wn)
        return f"g{bm}x{bn}{'w8' if waves == 8 else ''}{'d' if stages == 3 else ''}{'sk' if mode == 1 else ''}"
    x3 = re.search(r"conv_wino_x3<(\d+), (?:(\d+), )?(true|false)>", name)
    if x3:  # split-3 Winograd F(2x2,3x3): bench.py's key "w<tiles>x<couts>_bf16x3" (<MB, NBK, GENERAL> since round 5; <MB, GENERAL> before)
        return f"w{32 * int(x3.group(1))}x{32 * int(x3.group(2) or 2)}_bf16x3"
    if "conv_wino4_x3" in name:    # split-3 Winograd F(4x4,3x3)
        return "w4_32x64_bf16x3"
    xp = re.search(r"conv_igemm_x3p<(\d+)>", name)
    if xp:  # pipelined split-3 implicit GEMM: "128x<BN>p_bf16x3"
        return f"128x{int(xp.group(1))}p_bf16x3"
    if "conv_wino4_f32" in name:   # Winograd F(4x4,3x3): bench.py's key "w4_32x64"
        return "w4_32x64"
    if "conv_wino_f32_h" in name:
        return "w32x64h"
    if "conv_wino_f32_q" in name:
        return "w32x32q"
    wn = re.search(r"conv_wino_f32<(\d+), (\d+)>", name)
    if wn:  # Winograd F(2x2,3x3): bench.py's key "w<tiles>x<couts>" per workgroup
        return f"w{32 * int(wn.group(1))}x{32 * int(wn.group(2))}"
    b = re.search(r"conv_igemm_bf16<(\d+), (\d+), (\d+), (\d+)>", name)
    if b:   # AMP kernels: bench.py's key "<BM>x<BN>[w8]_bf16"
        bm, bn, wm, wn = (int(b.group(i)) for i in range(1, 5))
        return f"{bm}x{bn}{'w8' if (bm // wm) * (bn // wn) == 8 else ''}_bf16"
    m = re.search(r"conv_igemm_f32<(\d+), (\d+), (\d+), (\d+), (true|false)(?:, (true|false|\d))?>", name)
    if not m:
        return None
    bm, bn, wm, wn = (int(m.group(i)) for i in range(1, 5))
    waves = (bm // wm) * (bn // wn)
    return f"{bm}x{bn}{'w8' if waves == 8 else ''}{'d' if m.group(5) == 'true' else ''}{'sk' if m.group(6) in ('true', '1') else ''}{'p' if m.group(6) == '2' else ''}"


def calib(d, counter, kernel_sub, known_bytes):
    v = [r[4] for r in rows(d) if r[3] == counter and kernel_sub in r[0] and r[4] > 1024]
    return known_bytes / (sum(v) / len(v) * 1024.0), len(v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True, nargs="+", help="one or more FETCH_SIZE pass directories (several bench modes are merged)")
    ap.add_argument("--write", required=True, nargs="+")
    ap.add_argument("--calib-fetch")
    ap.add_argument("--calib-write")
    ap.add_argument("--calib-bytes", type=int, default=576 << 20)
    ap.add_argument("--merge", help="an earlier output of this tool: its (kernel, workgroups) entries are kept where these passes have none, "
                                    "and its calibration is used when no calibration pass is given")
    ap.add_argument("-o", "--out", required=True)
    a = ap.parse_args()
    fscale, wscale, cal = 2.0, 1.0, {"note": "guide defaults (no calibration run given)"}
    if a.calib_fetch and a.calib_write:
        f1, n1 = calib(a.calib_fetch, "FETCH_SIZE", "copyBuffer", a.calib_bytes)
        f2, n2 = calib(a.calib_fetch, "FETCH_SIZE", "reduce_kernel", a.calib_bytes)
        w1, n3 = calib(a.calib_write, "WRITE_SIZE", "copyBuffer", a.calib_bytes)
        w2, n4 = calib(a.calib_write, "WRITE_SIZE", "fillBufferAligned", a.calib_bytes)
        fscale, wscale = round((f1 + f2) / 2, 3), round((w1 + w2) / 2, 3)
        try:     # dword-per-lane reads in 32-byte runs (the Winograd input gather's pattern): tools/pmc_calib.py's strided copy
            f3, n5 = calib(a.calib_fetch, "FETCH_SIZE", "elementwise_kernel", a.calib_bytes)
        except ZeroDivisionError:
            f3, n5 = None, 0
        cal = {"known_bytes": a.calib_bytes, "fetch_scale_copy": round(f1, 4), "fetch_scale_reduce": round(f2, 4),
               "fetch_scale_dword_gather": (round(f3, 4) if f3 else None), "dword_gather_launches": n5,
               "dword_gather_note": "the strided copy of tools/pmc_calib.py re-fetches every line in 8 far-apart sweeps: 0.25 here = the streaming "
                                    "factor 2.0 / 8 re-fetches, i.e. the x 2 correction holds for dword gathers",
               "write_scale_copy": round(w1, 4), "write_scale_fill": round(w2, 4), "launches": [n1, n2, n3, n4],
               "note": "true bytes / (counter KiB x 1024) on tools/pmc_calib.py (576 MiB streaming copy / sum / fill)"}
    old = json.load(open(a.merge)) if a.merge else None
    if old and not (a.calib_fetch and a.calib_write):
        fscale, wscale, cal = old["fetch_scale"], old["write_scale"], old["calibration"]
    acc = defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
    for d in list(a.fetch) + list(a.write):
        last = None  # the stream-K launch a following conv_fixup_f32 dispatch belongs to
        for name, grid, wg, ctr, val in rows(d):
            if ctr not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            if "conv_fixup_f32" in name and last is not None:
                last[-1] += val   # bench.py times (and counts) the GEMM + its fix-up as one launch
                continue
            k = conv_key(name)
            if k:
                acc[(k, grid // wg)][ctr].append(val)
                last = acc[(k, grid // wg)][ctr] if k.endswith("sk") else None
    per = defaultdict(dict)
    for (k, wgs), v in sorted(acc.items()):
        if not v["FETCH_SIZE"] or not v["WRITE_SIZE"]:
            continue
        fb = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]) * 1024 * fscale
        wb = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"]) * 1024 * wscale
        per[k][str(wgs)] = {"fetch_bytes": round(fb), "write_bytes": round(wb), "launches": len(v["FETCH_SIZE"])}
    kept = 0
    if old:
        for k, tab in old["per_kernel"].items():
            for wgs, e in tab.items():
                if wgs not in per[k]:
                    per[k][wgs] = e
                    kept += 1
        print(f"{kept} entries kept from {a.merge}")
    json.dump({"counters": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes), KiB",
               "fetch_scale": fscale, "write_scale": wscale, "calibration": cal,
               "caveat": "TCC_EA (L2 memory-side) requests: Infinity-Cache hits are included, so this is an upper bound on HBM bytes",
               "per_kernel": per}, open(a.out, "w"), indent=1)
    print(f"fetch x{fscale} write x{wscale}; {sum(len(v) for v in per.values())} (kernel, grid) entries -> {a.out}")


if __name__ == "__main__":
    main()
