"""Host-side mirror of the reference's ORBVocabulary (include/ORBVocabulary.h:31 = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) over the C ABI:
only what the tracking path uses, transform(features, BowVector, FeatureVector, levelsup)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import Context, check, lib


class ORBVocabulary:
    def __init__(self, voc: dict, ctx: Context | None = None):
        """voc: dict(k, L, parent [n] i32, is_leaf [n] u8, desc [n,32] u8, weight [n] f64) - the rows of the vocabulary text file (ORBvoc.txt)."""
        self.L = lib()
        self.ctx = ctx or Context(0)
        a = dict(parent=np.ascontiguousarray(voc["parent"], np.int32), is_leaf=np.ascontiguousarray(voc["is_leaf"], np.uint8),
                 desc=np.ascontiguousarray(voc["desc"], np.uint8), weight=np.ascontiguousarray(voc["weight"], np.float64))
        h = C.c_void_p()
        check(self.L.planar_vocab_create(self.ctx.h, int(voc["k"]), int(voc["L"]), len(a["parent"]), a["parent"].ctypes.data, a["is_leaf"].ctypes.data,
                                         a["desc"].ctypes.data, a["weight"].ctypes.data, C.byref(h)))
        self.h = h
        self.n_words = check(self.L.planar_vocab_words(self.h))

    @classmethod
    def loadFromTextFile(cls, path: str, ctx: Context | None = None):
        """The reference's text format: 'k L scoring weighting', then per node 'parent isLeaf d0..d31 weight'."""
        with open(path) as f:
            k, L, scoring, weighting = (int(x) for x in f.readline().split()[:4])
            if (scoring, weighting) != (0, 0):
                raise ValueError("only L1_NORM scoring with TF_IDF weighting (ORBvoc.txt) is supported")
            rows = np.loadtxt(f, dtype=np.float64, ndmin=2)
        return cls(dict(k=k, L=L, parent=rows[:, 0].astype(np.int32), is_leaf=rows[:, 1].astype(np.uint8), desc=rows[:, 2:34].astype(np.uint8), weight=rows[:, 34]), ctx)

    def close(self):
        if getattr(self, "h", None):
            self.L.planar_vocab_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def transform(self, desc: np.ndarray, n=None, levelsup: int = 4):
        """desc [B,S,32] (or [S,32]) u8, n [B] -> dict(word, weight, node [B,S]; bow_word, bow_value [B,S], bow_n [B])."""
        single = desc.ndim == 2
        d = np.ascontiguousarray(desc[None] if single else desc, np.uint8)
        B, S = d.shape[:2]
        nn = np.full(B, S, np.int32) if n is None else np.ascontiguousarray(n, np.int32)
        out = dict(word=np.zeros((B, S), np.int32), weight=np.zeros((B, S)), node=np.zeros((B, S), np.int32), bow_word=np.zeros((B, S), np.int32), bow_value=np.zeros((B, S)),
                   bow_n=np.zeros(B, np.int32))
        check(self.L.planar_bow_transform(self.h, d.ctypes.data, nn.ctypes.data, B, S, levelsup, out["word"].ctypes.data, out["weight"].ctypes.data, out["node"].ctypes.data,
                                          out["bow_word"].ctypes.data, out["bow_value"].ctypes.data, out["bow_n"].ctypes.data))
        return out
