#!/usr/bin/env python3
"""Decode the reference's bundled image data into committed test fixtures.

Source (read-only, only present in the build container):
  /root/reference/data/ref_rect_{l,r}.zip  -- 640x480 rectified 8-bit stereo pair
  /root/reference/data/ref_xsbl_{l,r}.zip  -- the FPGA x-Sobel output of that pair (RTL golden,
                                              consumed by src/dvp/sim/sim_dvp.v:167-170,460-490)
Each .dat is text: one line per image row, 640 two-digit hex bytes separated by spaces, CRLF.

Output: tests/golden/ref_pair_640x480.npz with arrays rect_l, rect_r, xsbl_l, xsbl_r (uint8 480x640)
and tests/golden/ref_pair_640x480.sha256 (sha256 of each decoded array, for provenance).
These are data (inputs + reference-produced outputs), not reference source code.
"""
import hashlib
import io
import pathlib
import sys
import zipfile

import numpy as np

REF = pathlib.Path("/root/reference/data")
OUT = pathlib.Path(__file__).resolve().parents[1] / "tests" / "golden"


def decode(name):
    with zipfile.ZipFile(REF / f"{name}.zip") as z:
        (member,) = z.namelist()
        text = z.read(member).decode("ascii")
    rows = [[int(tok, 16) for tok in line.split()] for line in text.splitlines() if line.strip()]
    arr = np.array(rows, dtype=np.uint8)
    assert arr.shape == (480, 640), arr.shape
    return arr


def main():
    if not REF.exists():
        sys.exit("reference data not present; fixtures are already committed under tests/golden/")
    arrays = {k: decode(f"ref_{k}") for k in ("rect_l", "rect_r", "xsbl_l", "xsbl_r")}
    OUT.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT / "ref_pair_640x480.npz", **arrays)
    with open(OUT / "ref_pair_640x480.sha256", "w") as f:
        for k, a in arrays.items():
            f.write(f"{hashlib.sha256(a.tobytes()).hexdigest()}  {k} {a.shape[0]}x{a.shape[1]} uint8\n")
    print({k: hashlib.sha256(a.tobytes()).hexdigest()[:16] for k, a in arrays.items()})


if __name__ == "__main__":
    main()
