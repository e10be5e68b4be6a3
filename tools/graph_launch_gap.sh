#!/bin/bash
# When does hipGraphLaunch return relative to the previous replay's last kernel and the next replay's first kernel?
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/glg; rm -rf $O; mkdir -p $O
rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $O/prof -o run -- python3 bench.py --no-miou --no-cpu-baseline --profile-steps 0 --steps 8 --warmup 3 > $O/bench.json 2> $O/bench.err
ls $O/prof/* > $O/files.txt
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
kt = glob.glob(O + "/prof/**/*kernel_trace.csv", recursive=True)[0]
ht = [f for f in glob.glob(O + "/prof/**/*hip_api_trace.csv", recursive=True)][0]
ks = sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Start_Timestamp"]))
hs = [r for r in csv.DictReader(open(ht)) if r["Function"] == "hipGraphLaunch"]
out = open(O + "/summary.txt", "w")
print(f"{len(hs)} hipGraphLaunch calls; columns: call begin, call end, first kernel after call end ... (us, relative to call begin)", file=out)
adam = [i for i, r in enumerate(ks) if "adamw_flat_kernel" in r["Kernel_Name"]]
import bisect
starts = [int(r["Start_Timestamp"]) for r in ks]
ends = [int(r["End_Timestamp"]) for r in ks]
for h in hs[-6:]:
    b, e = int(h["Start_Timestamp"]), int(h["End_Timestamp"])
    # the last kernel that ENDED before this graph's first own kernel: find the big gap after the call
    i = bisect.bisect_left(starts, b)
    # kernels running/queued around: last kernel end before the largest gap within the next 40 ms
    j, best = i, (0, None)
    prev_end = max(ends[:i]) if i else b
    for k in range(i, min(i + 4000, len(ks))):
        gap = starts[k] - prev_end
        if gap > best[0]:
            best = (gap, k)
        prev_end = max(prev_end, ends[k])
        if starts[k] - b > 45e6:
            break
    g, k = best
    print(f"call {(e - b) / 1e3:9.1f} us long; GPU busy until {(max(ends[:k]) - b) / 1e3:10.1f} us after call begin; largest gap {g / 1e3:8.1f} us ends at {(starts[k] - b) / 1e3:10.1f} us "
          f"({ks[k - 1]['Kernel_Name'][:40]} -> {ks[k]['Kernel_Name'][:40]})", file=out)
out.close()
PY
rm -rf $O/prof
cat $O/summary.txt
