"""Generates tests/golden/ref_lcm.json from the REFERENCE's own lcm-gen generated message classes.

Run in the build container only (needs /root/reference):  make -C oracle ref && python tests/golden/make_ref_lcm.py
oracle/_ref/libref_lcm.so is lcm_msg/include/lcm_msg/{low_cmd_t,low_state_t,full_state_t}.hpp compiled in place
(oracle/Makefile); the hash constants are read from those headers' _computeHash bodies as well, so that the fixture pins
both the bytes and the fingerprints to the reference.
"""
import ctypes as C
import json
import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
REF = Path("/root/reference/lcm_msg/include/lcm_msg")
lib = C.CDLL(str(ROOT / "oracle/_ref/libref_lcm.so"))
lib.ref_hash.restype = C.c_longlong
NAMES = ["low_cmd_t", "low_state_t", "full_state_t"]
NF = [60, 40, 56]
ENC = [lib.ref_low_cmd_encode, lib.ref_low_state_encode, lib.ref_full_state_encode]


def special_values(rng, n):
    v = rng.normal(size=n) * 10.0 ** rng.integers(-3, 4, size=n)
    v[::7] = 0.0
    v[3::11] = -0.0
    v[5::13] = np.array([np.pi, -1e-300, 1e300, np.nextafter(1.0, 2.0)] * 8)[: len(v[5::13])]
    return v


def main():
    rng = np.random.default_rng(20260924)
    doc = {"source": "lcm_msg/include/lcm_msg/*.hpp compiled in place (oracle/Makefile: ref)", "types": []}
    for t, name in enumerate(NAMES):
        src = (REF / f"{name}.hpp").read_text()
        base = int(re.search(r"uint64_t hash = 0x([0-9a-f]{16})LL;", src).group(1), 16)
        msgs = []
        for k in range(6):
            ts = int(rng.integers(-2**62, 2**62)) if k else 1726000000123456789
            f = special_values(rng, NF[t])
            buf = (C.c_ubyte * 1024)()
            n = ENC[t](C.c_longlong(ts), f.ctypes.data_as(C.POINTER(C.c_double)), buf, 1024)
            assert n == lib.ref_size(t)
            msgs.append({"timestamp": ts, "fields_hex": f.tobytes().hex(), "bytes_hex": bytes(buf[:n]).hex()})
        doc["types"].append({"name": name, "hash_constant_in_header": f"{base:016x}", "fingerprint": f"{lib.ref_hash(t) & (2**64 - 1):016x}",
                             "encoded_size": lib.ref_size(t), "n_fields": NF[t], "messages": msgs})
    out = ROOT / "tests/golden/ref_lcm.json"
    out.write_text(json.dumps(doc, indent=1))
    print("wrote", out, out.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
