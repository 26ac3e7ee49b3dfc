"""Timing diagnosis only: bench.py's step with some extraction stages LEFT OUT, to read a stage's marginal cost off the step (DESIGN.md §4's tables).

    python tools/stage_skip.py lsd,planepost -- --steps 8 --warmup 3 --cpu-seconds 0 --latency-reps 0 --pcie-steps 0

Stages: lsd (both halves of ExtractLineSegment; isLineGood stays), peac, planepost (voxel clouds + refit), normals, orb.  The product has no such switch
(round 4's PLANAR_TRACK_SKIP environment variable is gone): this file subclasses TrackPipeline, overrides the stage methods with no-ops, hands the subclass to
bench.main() and relabels the printed line - its results are meaningless as tracking output (the skipped stages' buffers hold zeros) and its `metric` says so."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    if "--" not in sys.argv:
        raise SystemExit(__doc__)
    cut = sys.argv.index("--")
    skip = set(x for x in ",".join(sys.argv[1:cut]).split(",") if x)
    known = {"lsd", "peac", "planepost", "normals", "orb"}
    if not skip or skip - known:
        raise SystemExit(f"stages to skip: a comma list out of {sorted(known)}")
    from planarslam_amd import track
    from planarslam_amd._lib import check

    class SkippingPipeline(track.TrackPipeline):
        def _lines_head(self, k, gray):
            if "lsd" not in skip: super()._lines_head(k, gray)

        def _lines_tail(self, i, k, depth):
            if "lsd" not in skip:
                return super()._lines_tail(i, k, depth)
            # isLineGood still runs (on the zero key lines the skipped detector left), as round 4's switch did
            import numpy as np
            L, B, l3, c = self.L, self.B, self.l3[k], self.cam
            check(L.planar_is_line_good_dev(self.ctx_lsds[k].h, B, self.kls[k].data_ptr(), self.nl[k].data_ptr(), 40, depth.data_ptr(), self.W, self.H, self.W, self.W * self.H,
                                            float(np.float32(1.0 / 5000.0)), c["fx"], c["fy"], c["cx"], c["cy"], l3["seeds"].data_ptr(), l3["depth_line"].data_ptr(),
                                            l3["lines3d"].data_ptr(), l3["good"].data_ptr(), l3["direction"].data_ptr(), l3["n_inliers"].data_ptr(), l3["packed"].data_ptr(),
                                            l3["n_good"].data_ptr()))

        def _planes(self, k, depth):
            if "peac" not in skip: super()._planes(k, depth)

        def _plane_clouds(self, k, depth):
            if "planepost" not in skip: super()._plane_clouds(k, depth)

        def _normals(self, k, depth):
            if "normals" not in skip: super()._normals(k, depth)

        def _points(self, k, gray):
            if "orb" not in skip: super()._points(k, gray)

    track.TrackPipeline = SkippingPipeline
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[cut + 1:]
    import bench
    bench.PARITY_SAMPLE = False            # nothing to compare: stages are missing
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    for line in buf.getvalue().splitlines():
        try:
            d = json.loads(line)
        except ValueError:
            print(line); continue
        d["metric"] = "TIMING DIAGNOSIS, NOT A BENCH LINE: step without " + ",".join(sorted(skip))
        d["stages_skipped"] = sorted(skip)
        print(json.dumps(d))


if __name__ == "__main__":
    main()
