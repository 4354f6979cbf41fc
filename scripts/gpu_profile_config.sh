#!/bin/bash
# Evidence for ONE bench configuration: bench line, rocprofv3 kernel stats of the same command, HBM traffic from two
# separate PMC passes (FETCH_SIZE, WRITE_SIZE) summarised per launch, VALUBusy / VALUUtilization of the trace kernel.
# usage (on the box, from the repo root): bash scripts/gpu_profile_config.sh <name> [bench args]
name=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
# (1) the plain line of this box, for the record; (2) the SAME command --lean under rocprofv3 --kernel-trace --stats: its own
# JSON line (HIP events in that very process) + the per-launch trace, reduced to the timed region's launches
# (scripts/timed_region_stats.py) -- the two must agree to 1 % (VERDICT r03 task 3)
timeout 900 python3 $R/bench.py "$@" 2>/dev/null | tail -1 > $O/prof_${name}_bench_plain.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$name /tmp/pf_$name /tmp/pw_$name
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$name -- python3 $R/bench.py --lean "$@" > $O/prof_${name}_rocprof.log 2>&1
grep '"metric"' $O/prof_${name}_rocprof.log | tail -1 > $O/prof_${name}_bench.json
for f in $(find /tmp/kt_$name -name "*kernel_stats.csv"); do cp $f $O/prof_${name}_kernel_stats.csv; done
for f in $(find /tmp/kt_$name -name "*kernel_trace.csv"); do
  python3 $R/scripts/timed_region_stats.py $f $O/prof_${name}_bench.json $O/prof_${name}_timed_region.json trace_ 8 1
done
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$name -- python3 $R/bench.py --steps 3 --warmup 1 --ramp-seconds 0 --lean "$@" >> $O/prof_${name}_rocprof.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw_$name -- python3 $R/bench.py --steps 3 --warmup 1 --ramp-seconds 0 --lean "$@" >> $O/prof_${name}_rocprof.log 2>&1
for f in $(find /tmp/pf_$name -name "*counter_collection.csv"); do head -1 $f > $O/prof_${name}_pmc_fetch.csv; grep "trace_" $f >> $O/prof_${name}_pmc_fetch.csv; done
for f in $(find /tmp/pw_$name -name "*counter_collection.csv"); do head -1 $f > $O/prof_${name}_pmc_write.csv; grep "trace_" $f >> $O/prof_${name}_pmc_write.csv; done
python3 $R/scripts/summarize_pmc.py $O/prof_${name}_pmc_fetch.csv $O/prof_${name}_pmc_write.csv $O/prof_${name}_pmc_summary.json trace_ > /dev/null
for ctr in VALUBusy VALUUtilization SQ_INSTS_VALU; do
  rm -rf /tmp/dv_$name
  timeout 900 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/dv_$name -- python3 $R/bench.py --steps 3 --warmup 1 --ramp-seconds 0 --lean "$@" >> $O/prof_${name}_rocprof.log 2>&1
  for f in $(find /tmp/dv_$name -name "*counter_collection.csv"); do head -1 $f > $O/prof_${name}_$ctr.csv; grep "trace_" $f >> $O/prof_${name}_$ctr.csv; done
done
python3 - <<PY
import csv, json
name = "$name"; O = "$O"
b = json.load(open(f"{O}/prof_{name}_bench.json"))
print(name, "bench: ms/step %.3f value %.0f frac %.3f kernel_ms %.3f" % (b["ms_per_step"], b["value"], b["roofline"]["frac"], b["roofline"]["kernel_ms"]))
for r in csv.DictReader(open(f"{O}/prof_{name}_kernel_stats.csv")):
    if float(r["Percentage"]) > 1.0:
        print("   %-60s calls %5s avg %10.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
s = json.load(open(f"{O}/prof_{name}_pmc_summary.json"))
print("   HBM bytes per trace launch %.1f MB (fetch x2 %.1f + write %.1f)" % (s.get("hbm_bytes_per_launch", 0) / 1e6, (s["fetch_bytes_corrected"] or 0) / 1e6, (s["write_bytes"] or 0) / 1e6))
try:   # wave-level VALU instructions per trace launch -> into the PMC summary (bench.py reports it per 64 ray-steps)
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f"{O}/prof_{name}_SQ_INSTS_VALU.csv")) if r["Counter_Name"] == "SQ_INSTS_VALU"]
    s["valu_insts_per_launch"] = sum(v) / len(v)
    s["valu_insts_launches"] = len(v)
    json.dump(s, open(f"{O}/prof_{name}_pmc_summary.json", "w"), indent=1)
    rs = b["roofline"]["ray_steps_per_launch"]
    print("   SQ_INSTS_VALU %.4g per launch = %.1f per 64 ray-steps" % (s["valu_insts_per_launch"], s["valu_insts_per_launch"] * 64 / rs))
except Exception as e:
    print("   SQ_INSTS_VALU failed", e)
for ctr in ("VALUBusy", "VALUUtilization"):
    try:
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f"{O}/prof_{name}_{ctr}.csv")) if r["Counter_Name"] == ctr]
        print("   %s mean %.1f over %d launches" % (ctr, sum(v) / len(v), len(v)))
    except Exception as e:
        print("  ", ctr, "failed", e)
PY
