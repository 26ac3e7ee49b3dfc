"""tests/golden/bow_ref.npz: outputs of the REAL DBoW2 (oracle/_ref/ref_bow = Thirdparty/DBoW2 compiled where it lies) on the seeded cases of tests/bow_cases.py."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bow_cases as cases  # noqa: E402
import oracle_lib as O  # noqa: E402
from planarslam_amd import synth  # noqa: E402

out = {}
with tempfile.TemporaryDirectory() as d:
    for name in cases.CASES:
        voc, q, levelsup = cases.build(name)
        path = os.path.join(d, name + ".txt")
        synth.write_vocabulary_text(voc, path)
        r = O.run_ref_bow(path, q, levelsup)
        for k, v in r.items():
            out[f"{name}/{k}"] = v
        print(name, len(r["bow_word"]), "words,", int((r["node"] < 0).sum()), "stopped features")
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bow_ref.npz"), **out)
