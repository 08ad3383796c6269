#!/usr/bin/env python3
"""The INPUTS of BASELINE.json config 3 as data: the 32 demo_data families the reference ships (msa, tree and contact-map
text files of /root/reference/demo_data), packed into tests/golden/demo32_co_inputs.npz so that the GPU test can run this
package's own counting chain (maximal matching -> cb_count_co_transitions) on all of them and compare with the counts the
REFERENCE produced from the same files (coevo_demo_full.npz, made by make_golden_s400_full.py: sum C = 1 057 194,
730 864 non-zeros).  Data files only -- nothing of the reference's source.  Build container only.
Usage: python tests/golden/make_golden_demo32_inputs.py"""
import os

import numpy as np

REF = "/root/reference/demo_data"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    fams = sorted(f[:-4] for f in os.listdir(os.path.join(REF, "msas")) if f.endswith(".txt"))
    out = {"families": np.array(fams)}
    for kind, sub in (("msa", "msas"), ("tree", "trees"), ("contact_map", "contact_maps")):
        blobs = [open(os.path.join(REF, sub, f + ".txt"), "rb").read() for f in fams]
        out[f"{kind}_bytes"] = np.frombuffer(b"".join(blobs), dtype=np.uint8)      # the files' bytes, back to back
        out[f"{kind}_offsets"] = np.cumsum([0] + [len(b) for b in blobs]).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "demo32_co_inputs.npz"), **out)
    print(len(fams), "families ->", os.path.getsize(os.path.join(HERE, "demo32_co_inputs.npz")), "bytes")


if __name__ == "__main__":
    main()
