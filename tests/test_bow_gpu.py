"""DBoW2 vocabulary transform on the GPU (planar_bow_transform) vs what the REAL vendored DBoW2 returned on the same seeded vocabularies
(tests/golden/bow_ref.npz, tools/gen_golden_bow.py) and vs the oracle: word ids, FeatureVector nodes and the FP64 BowVector, every bit."""
import os

import numpy as np
import pytest

import bow_cases as cases
import oracle_lib as O
from planarslam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from planarslam_amd._lib import Context
    return Context(0)


@pytest.mark.parametrize("name", list(cases.CASES))
def test_bow_hip_equals_real_dbow2_fixture(ctx, golden_dir, name):
    from planarslam_amd.bow import ORBVocabulary
    g = np.load(os.path.join(golden_dir, "bow_ref.npz"))
    voc, q, levelsup = cases.build(name)
    r = ORBVocabulary(voc, ctx).transform(q, levelsup=levelsup)
    n = len(q)
    np.testing.assert_array_equal(r["word"][0, :n], g[f"{name}/word"])
    np.testing.assert_array_equal(r["weight"][0, :n], g[f"{name}/weight"])
    np.testing.assert_array_equal(r["node"][0, :n], g[f"{name}/node"])
    m = int(r["bow_n"][0])
    assert m == len(g[f"{name}/bow_word"])
    np.testing.assert_array_equal(r["bow_word"][0, :m], g[f"{name}/bow_word"])
    np.testing.assert_array_equal(r["bow_value"][0, :m], g[f"{name}/bow_value"])


def test_bow_batch_ragged_and_text_file_loader(ctx, tmp_path):
    from planarslam_amd.bow import ORBVocabulary
    voc = synth.vocabulary(k=10, L=4, seed=301)
    path = str(tmp_path / "voc.txt")
    synth.write_vocabulary_text(voc, path)
    V = ORBVocabulary.loadFromTextFile(path, ctx)
    assert V.n_words == 10 ** 4
    B, S = 5, 1500
    desc = np.stack([np.concatenate([synth.vocabulary_queries(voc, 1200, 40 + b), np.zeros((S - 1200, 32), np.uint8)]) for b in range(B)])
    n = np.array([1200, 0, 1, 777, 1200], np.int32)
    r = V.transform(desc, n, levelsup=2)
    orc = O.VocabOracle(voc)
    for b in range(B):
        w = orc.transform(desc[b, :n[b]], 2)
        k = int(n[b])
        for key in ("word", "weight", "node"):
            np.testing.assert_array_equal(r[key][b, :k], w[key], err_msg=f"{key} frame {b}")
        m = int(r["bow_n"][b])
        assert m == len(w["bow_word"])
        np.testing.assert_array_equal(r["bow_word"][b, :m], w["bow_word"]); np.testing.assert_array_equal(r["bow_value"][b, :m], w["bow_value"])
        assert (r["node"][b, k:] == -1).all()


def test_bow_nodes_feed_search_by_bow(ctx):
    """transform -> SearchByBoW on the device-format FeatureVectors (one node id per feature), against the oracle chain."""
    from planarslam_amd.bow import ORBVocabulary
    from planarslam_amd.guided import ORBmatcher
    voc = synth.vocabulary(k=10, L=4, seed=302)
    V = ORBVocabulary(voc, ctx)
    kf, f = synth.guided_bow(B=2, N=800, seed=303)
    rk, rf = V.transform(kf["desc"], kf["n"], levelsup=2), V.transform(f["desc"], f["n"], levelsup=2)
    kf2, f2 = dict(kf, node=rk["node"]), dict(f, node=rf["node"])
    m, nm = ORBmatcher(0.7, True, ctx).SearchByBoW(kf2, f2)
    orc = O.VocabOracle(voc)
    for b in range(2):
        kf2["node"][b, :kf["n"][b]] = orc.transform(kf["desc"][b, :kf["n"][b]], 2)["node"]
        f2["node"][b, :f["n"][b]] = orc.transform(f["desc"][b, :f["n"][b]], 2)["node"]
    wm, wn = O.search_by_bow(kf2, f2, 0.7)
    np.testing.assert_array_equal(m, wm); np.testing.assert_array_equal(nm, wn)
    assert nm.min() > 50
