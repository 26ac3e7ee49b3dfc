"""Pins oracle/bow_oracle.cpp (DBoW2 vocabulary transform) to the REAL vendored DBoW2: tests/golden/bow_ref.npz holds what
Thirdparty/DBoW2's TemplatedVocabulary::transform returned (word ids, idf weights, FeatureVector nodes, the L1-normalised BowVector) on the seeded
vocabularies of tests/bow_cases.py, loaded through its own loadFromTextFile; the oracle must give the same bits.  With the binary present it is also run live."""
import os

import numpy as np
import pytest

import bow_cases as cases
import oracle_lib as O
from planarslam_amd import synth


@pytest.mark.parametrize("name", list(cases.CASES))
def test_bow_oracle_equals_real_dbow2_fixture(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "bow_ref.npz"))
    voc, q, levelsup = cases.build(name)
    r = O.VocabOracle(voc).transform(q, levelsup)
    for k in ("word", "node", "bow_word"):
        np.testing.assert_array_equal(r[k], g[f"{name}/{k}"], err_msg=k)
    for k in ("weight", "bow_value"):
        np.testing.assert_array_equal(r[k], g[f"{name}/{k}"], err_msg=k)       # FP64, same operation order: every bit
    if len(r["bow_value"]):
        assert abs(r["bow_value"].sum() - 1.0) < 1e-12 and (np.diff(r["bow_word"]) > 0).all()


@pytest.mark.skipif(not os.path.exists(O.ref_bow_path()), reason="oracle/_ref/ref_bow not built (reference tree absent)")
def test_bow_oracle_equals_real_dbow2_live(tmp_path):
    voc = synth.vocabulary(k=10, L=3, seed=123)
    q = synth.vocabulary_queries(voc, 500, 11)
    path = str(tmp_path / "voc.txt")
    synth.write_vocabulary_text(voc, path)
    for levelsup in (0, 1, 2, 5):
        ref, got = O.run_ref_bow(path, q, levelsup), O.VocabOracle(voc).transform(q, levelsup)
        for k in ref:
            np.testing.assert_array_equal(got[k], ref[k], err_msg=f"{k} levelsup={levelsup}")
