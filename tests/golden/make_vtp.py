"""Generates the committed .vtp fixtures (feature files in the reference's on-disk format, written by the
pure-Python writer lrr_amd/vtp.py) and the arrays they must read back as.

    python tests/golden/make_vtp.py

The reference ships no data files; these are synthetic clouds of lrr_amd/synth.py.  One file per
encoding family vtkXMLPolyDataWriter can produce.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from lrr_amd import synth, vtp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [  # name, format, compress, header_type
    ("features_ascii", "ascii", False, "UInt32"),
    ("features_binary", "binary", False, "UInt32"),
    ("features_appended_zlib", "appended-base64", True, "UInt32"),
    ("features_appended_raw64", "appended-raw", False, "UInt64"),
]


def cloud():
    pair = synth.make_pair(400, 400, 424242, kind="affine")
    return pair.fixed


def main():
    fs = cloud()
    arrays = vtp.disk_arrays(fs)
    for name, fmt, comp, ht in CASES:
        vtp.write_vtp(os.path.join(HERE, name + ".vtp"), arrays, fmt=fmt, compress=comp, header_type=ht)
    # what every file must read back as: float32 on disk, widened to double
    np.savez_compressed(os.path.join(HERE, "features_vtp_expected.npz"),
                        **{k: np.asarray(v, dtype=np.float32 if k != "shapes" else np.uint8).astype(np.float64) for k, v in arrays.items()})


if __name__ == "__main__":
    main()
