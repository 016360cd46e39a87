#!/bin/bash
# where the cost of a pass goes: per-kernel totals (rocprofv3 --kernel-trace --stats) of four frames of C4 at 13 passes per frame (48 GiB of
# job buffers, the library default) and at 4 (160 GiB, what bench.py sets)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r04_passcost; mkdir -p $OUT
for mb in 49152 163840; do
  FRAMES=3 PTCORE_L_BUDGET_MB=$mb rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b$mb -- python3 tools/r04_pt_render_default.py --keep-env > $OUT/b$mb.json 2> $OUT/b$mb.err || exit 1
  f=$(find $OUT/b$mb -name '*_kernel_stats.csv' | head -1)
  echo "== budget $mb MiB"; cat $OUT/b$mb.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('steady %.1f ms, %d passes' % (d['steady_frame_ms_median'], d['passes_per_frame']))"
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-60s calls %5s  total %9.2f ms  avg %8.3f ms  %5s %%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, r["Percentage"]))
PY
done
