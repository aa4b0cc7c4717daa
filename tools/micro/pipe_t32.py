"""Debug probe: run frames through FramePipeline with the 32 x 64 conv_wino_x3 tile forced on the small maps and compare EVERY
pipelined frame with the single-stream result of the same engine.  python tools/micro/pipe_t32.py [where2com|cobevt|v2xvit] [agents] [frames]"""
import os
import sys
from types import SimpleNamespace

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench                                                                       # noqa: E402
from airv2x_perception_amd.opencood_iface.engine import FramePipeline, Where2ComEngine   # noqa: E402


def main():
    model_name = sys.argv[1] if len(sys.argv) > 1 else "where2com"
    agents = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    t32 = os.environ.get("T32", "1") == "1"
    dev = torch.device("cuda", 0)
    a = SimpleNamespace(model=model_name, amp=False, gemm="x3", agents=agents, points=8192, mods=("lidar",))
    hy, args, dd, clouds, types = bench.build_inputs(agents, 8192, dev, only=None, model=model_name, modalities=("lidar",))
    Where2ComEngine.WINO_X3_T32 = t32
    count = {32: 0, 64: 0}
    orig = Where2ComEngine.wino_x3_tile.__func__

    def counted(cls, L, h, w):
        t = orig(cls, L, h, w)
        count[(t >> 16) & 0xff] += 1
        return t
    Where2ComEngine.wino_x3_tile = classmethod(counted)
    guards = {}
    if os.environ.get("GUARD") == "1":      # every workspace buffer between two 1 MiB sentinel bands: out-of-bounds WRITES of any kernel of the frame
        import math
        G = 1 << 20

        def gbuf(self, name, shape, dtype=torch.float32):
            key = (name, tuple(shape), dtype)
            t = self.ws.get(key)
            if t is None:
                n = math.prod(shape) * torch.empty((), dtype=dtype).element_size()
                raw = torch.full((n + 2 * G,), 0xA5, dtype=torch.uint8, device=self.device)
                t = raw[G:G + n].view(dtype).view(tuple(shape))
                guards[(id(self),) + key] = (raw, n)
                self.ws[key] = t
                self._ws_bytes += n
            self._ws_used[key] = self._frame
            return t
        Where2ComEngine.buf = gbuf
    model, eng, sd = bench.make_model(a, args, dev)
    if os.environ.get("WX3") == "0":
        eng.wino_x3 = False
    if os.environ.get("X3P") == "0":
        eng.x3p = False
    print("engine: wino_x3", eng.wino_x3, "x3p", eng.x3p, flush=True)
    keys = ("psm", "rm", "obj") if model_name != "v2xvit" else ("psm", "rm")
    out = model(dd)
    torch.cuda.synchronize()
    keys = [k for k in keys if k in out]
    ref = {k: out[k].clone() for k in keys}
    out = model(dd)
    torch.cuda.synchronize()
    print("single stream repeat equal:", all(torch.equal(out[k], ref[k]) for k in keys), flush=True)
    if guards:
        G = 1 << 20
        nbad = 0
        for key, (raw, n) in guards.items():
            lo, hi = raw[:G] != 0xA5, raw[G + n:] != 0xA5
            if bool(lo.any()) or bool(hi.any()):
                nbad += 1
                li = lo.nonzero().reshape(-1)
                hi_i = hi.nonzero().reshape(-1)
                print("  GUARD BROKEN around", key[1:], "bytes below:", int(lo.sum()), "(last at -%d)" % (G - int(li[-1])) if len(li) else "",
                      "bytes above:", int(hi.sum()), "(first at +%d, last at +%d)" % (int(hi_i[0]), int(hi_i[-1])) if len(hi_i) else "", flush=True)
        print(f"guard bands checked around {len(guards)} workspace buffers: {nbad} broken", flush=True)
    if os.environ.get("TRACE", "0") == "1":
        pipe = FramePipeline(eng, 3)
        recs = {}

        def instrument(e, ei):
            oc, ol = e.conv, e.ln

            def conv(L, x, n, h, w, out, **kw):
                si = x.float().sum(dtype=torch.float64)
                r = oc(L, x, n, h, w, out, **kw)
                recs[ei].append((f"conv {L.cin}->{L.cout} k{L.ks} n{n} {h}x{w} {kw.get('out_coff', 0)}", si, out.float().sum(dtype=torch.float64)))
                return r

            def ln(x, gb, y, nt, c):
                si = x.float().sum(dtype=torch.float64)
                r = ol(x, gb, y, nt, c)
                recs[ei].append((f"ln {nt}x{c}", si, y.float().sum(dtype=torch.float64)))
                return r
            e.conv, e.ln = conv, ln
        for ei, e in enumerate(pipe.engines):
            instrument(e, ei)
        for ei in range(3):
            recs[ei] = []
        for _ in range(3):
            pipe.submit(dd)
            torch.cuda.synchronize()
        serial = {ei: [(n, float(a), float(b)) for n, a, b in recs[ei]] for ei in recs}
        for rnd in range(2):
            for ei in range(3):
                recs[ei] = []
            for _ in range(3):
                pipe.submit(dd)
            torch.cuda.synchronize()
            for ei in range(3):
                cur = [(n, float(a), float(b)) for n, a, b in recs[ei]]
                assert len(cur) == len(serial[ei])
                firsts = [i for i, (c, s0) in enumerate(zip(cur, serial[ei])) if c != s0]
                print(f"round {rnd} engine {ei}: {len(firsts)} of {len(cur)} recorded ops differ; first:", flush=True)
                for i in firsts[:4]:
                    print("     op", i, cur[i][0], "input sum equal:", cur[i][1] == serial[ei][i][1], "output sum equal:", cur[i][2] == serial[ei][i][2],
                          "| previous op:", cur[i - 1][0] if i else None, flush=True)
        return
    if os.environ.get("WSDIFF", "0") == "1":
        pipe = FramePipeline(eng, 3)
        for _ in range(3):
            pipe.submit(dd)
            torch.cuda.synchronize()
        snaps = [{k: t.clone() for k, t in e.ws.items()} for e in pipe.engines]
        for rnd in range(3):
            for _ in range(3):
                pipe.submit(dd)
            torch.cuda.synchronize()
            for ei, e in enumerate(pipe.engines):
                diff = []
                for k, t in e.ws.items():
                    if k in snaps[ei] and not torch.equal(t, snaps[ei][k]):
                        a, b = t.float(), snaps[ei][k].float()
                        nbad = int((a != b).sum())
                        first = int((a != b).reshape(-1).nonzero()[0])
                        diff.append((k[0], tuple(k[1]), nbad, t.numel(), first, float((a - b).abs().max())))
                print(f"round {rnd} engine {ei}: {len(diff)} of {len(e.ws)} workspace buffers differ from the serial run", flush=True)
                for d in sorted(diff)[:60]:
                    print("     ", d, flush=True)
        return
    depth = int(os.environ.get("DEPTH", "3"))
    serial = os.environ.get("SERIAL", "0") == "1"
    pipe = FramePipeline(eng, depth)
    bad = 0
    outs = []
    sums = []
    for f in range(frames):
        po, ev = pipe.submit(dd)
        if serial:
            torch.cuda.synchronize()
            sums.append((f % depth, [float(po[k].double().sum()) for k in keys], all(torch.equal(po[k], ref[k]) for k in keys)))
        outs.append((po, ev))
        if len(outs) >= 3:
            o, e = outs.pop(0)
            e.synchronize()
            if not all(torch.equal(o[k], ref[k]) for k in keys):
                bad += 1
                print("  frame", f - 2, "differs:", {k: float((o[k] - ref[k]).abs().max()) for k in keys}, flush=True)
    for o, e in outs:
        e.synchronize()
        if not all(torch.equal(o[k], ref[k]) for k in keys):
            bad += 1
    if serial:
        for x in sums[:8]:
            print("   slot, sums, equal:", x, flush=True)
    torch.cuda.synchronize()
    again = model(dd)
    torch.cuda.synchronize()
    print("single-stream frame AFTER the overlapped frames equals the first one:", all(torch.equal(again[k], ref[k]) for k in keys), flush=True)
    print(f"depth {depth} serial {serial} {model_name} agents {agents} t32 {t32}: {bad} / {frames} pipelined frames differ from the single-stream frame; conv_wino_x3 launches "
          f"by tile {count}", flush=True)


if __name__ == "__main__":
    main()
