"""Seeded vocabulary / query cases shared by tools/gen_golden_bow.py (real DBoW2 -> tests/golden/bow_ref.npz), the oracle test and the GPU test."""
from planarslam_amd import synth

# name -> (vocabulary kwargs, n queries, query seed, levelsup)
CASES = {
    "k10_L3": (dict(k=10, L=3, seed=77), 1000, 5, 2),
    "k10_L4_levelsup4": (dict(k=10, L=4, seed=78), 1200, 6, 4),        # levelsup >= L: every feature's node is the root (nid_level <= 0)
    "k10_L4_levelsup1": (dict(k=10, L=4, seed=78), 700, 7, 1),
    "k4_L5": (dict(k=4, L=5, seed=79, stop_frac=0.2), 900, 8, 3),      # many stopped words
    "few": (dict(k=10, L=3, seed=80), 3, 9, 2),
}


def build(name):
    kw, n, qseed, levelsup = CASES[name]
    voc = synth.vocabulary(**kw)
    return voc, synth.vocabulary_queries(voc, n, qseed), levelsup
