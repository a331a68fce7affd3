#!/usr/bin/env python3
"""Quick GPU parity + timing probe (development aid; the real checks live in tests/)."""
import ctypes, os, sys, time, random, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eccoxide_amd as E

ORA = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "..", "oracle", "liboracle.so"))

def oracle_var(cid, ks, pts, n, fb, ed=False):
    out = ctypes.create_string_buffer(n * 2 * fb); inf = ctypes.create_string_buffer(n)
    proj = ctypes.create_string_buffer(n * (4 if ed else 3) * fb)
    ORA.eccx_oracle_scalarmul_var(cid, ctypes.c_size_t(n), ks, pts, out, inf, proj, 16)
    return out.raw, inf.raw, proj.raw

def oracle_base(cid, ks, n, fb, ed=False):
    out = ctypes.create_string_buffer(n * 2 * fb); inf = ctypes.create_string_buffer(n)
    proj = ctypes.create_string_buffer(n * (4 if ed else 3) * fb)
    ORA.eccx_oracle_scalarmul_base(cid, ctypes.c_size_t(n), ks, out, inf, proj, 16)
    return out.raw, inf.raw, proj.raw

def main():
    eng = E.Engine(0)
    rng = random.Random(7)
    only = sys.argv[1:] or list(E.CURVE_IDS)
    for name in only:
        cid = E.CURVE_IDS[name]
        fb, sb = E.field_bytes(cid), E.scalar_bytes(cid)
        ed = cid == E.ED25519
        n = 300
        ks = bytes(rng.getrandbits(8) for _ in range(n * sb))
        if name == "p521r1":   # keep scalars < 2^521
            ks = b"".join(bytes([ks[i * sb] & 1]) + ks[i * sb + 1:(i + 1) * sb] for i in range(n))
        # bases: r*G from the oracle's fixed-base path
        rs = bytes(rng.getrandbits(8) for _ in range(n * sb))
        if name == "p521r1":
            rs = b"".join(bytes([rs[i * sb] & 1]) + rs[i * sb + 1:(i + 1) * sb] for i in range(n))
        pts, _, _ = oracle_base(cid, rs, n, fb, ed)
        t = time.time(); o_out, o_inf, o_proj = oracle_var(cid, ks, pts, n, fb, ed); t_or = time.time() - t
        t = time.time(); g_out, g_fl, g_proj = eng.scalarmul_var(cid, ks, pts, want_proj=True); t_g = time.time() - t
        print(json.dumps({"curve": name, "op": "var", "n": n, "affine_ok": g_out == o_out, "flags_ok": g_fl == o_inf,
                          "proj_ok": g_proj == o_proj, "oracle_s": round(t_or, 3), "gpu_s": round(t_g, 3)}), flush=True)
        o_out, o_inf, o_proj = oracle_base(cid, ks, n, fb, ed)
        t = time.time(); g_out, g_fl, g_proj = eng.scalarmul_base(cid, ks, want_proj=True); t_g = time.time() - t
        print(json.dumps({"curve": name, "op": "base", "n": n, "affine_ok": g_out == o_out, "flags_ok": g_fl == o_inf,
                          "proj_ok": g_proj == o_proj, "gpu_s": round(t_g, 3)}), flush=True)
    # timing, device resident
    import torch
    for name, n in (("p256r1", 1 << 16), ("p256r1", 1 << 20), ("ed25519", 1 << 20)):
        if name not in only: continue
        cid = E.CURVE_IDS[name]; fb, sb = E.field_bytes(cid), E.scalar_bytes(cid)
        g = torch.Generator(device="cpu"); g.manual_seed(1)
        ks = torch.randint(0, 256, (n, sb), dtype=torch.uint8, generator=g).cuda()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if name == "ed25519":
            out, fl = eng.scalarmul_base_t(cid, ks); torch.cuda.synchronize()
            e0.record(); out, fl = eng.scalarmul_base_t(cid, ks, out, fl); e1.record(); torch.cuda.synchronize()
        else:
            pts1, _ = eng.scalarmul_base_t(cid, ks); torch.cuda.synchronize()
            out, fl = eng.scalarmul_var_t(cid, ks, pts1); torch.cuda.synchronize()
            e0.record(); out, fl = eng.scalarmul_var_t(cid, ks, pts1, out, fl); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(json.dumps({"curve": name, "n": n, "ms": round(ms, 3), "ops_per_s": round(n / ms * 1e3)}), flush=True)

if __name__ == "__main__":
    main()
