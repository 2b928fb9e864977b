#!/bin/bash
# One GPU session = a list of tasks run in order on the gpurun box; every task writes gpurun_out/<tag>_<task>.txt.
#   gpurun --timeout 2400 -- 'bash tools/gpu_session.sh r06a tests fuzz:24:11 smoke bench harness:chain_harness ab:TN_CHAIN:0:1:3'
# tasks:  tests[:pytest -k expression]   full -m gpu suite (or a subset)
#         fuzz:N:SEED                    tools/fuzz_paths.py
#         smoke                          __graft_entry__.smoke()
#         bench[:extra bench.py flags]   the driver's command, one JSON line
#         harness:NAME[:args]            tools/NAME (a binary built from tools/NAME.hip by build())
#         ab:ENV:A:B:ROUNDS              same-box A/B of the headline step under ENV=A / ENV=B (tools/ab_env.sh)
#         legs[:ROUNDS]                  other_configs legs A/B of two libraries (tools/ab_legs.sh, ab_libs/)
#         profiles                       tools/collect_profiles.sh <tag> (kernel stats + PMC traffic -> gpurun_out/)
#         sq                             tools/collect_sq_counters.sh <tag>
#         py:SCRIPT[:args]               python tools/SCRIPT args
#         legab:LEG:ENV:ROUNDS:V1,V2,..  same-box A/B of ONE other_configs leg under ENV=V1 / V2 / .. (tools/env_ab.sh)
#         legprof:LEG                    kernel-time table of one other_configs leg (tools/prof_leg.sh)
#         gaps:LEG                       idle time between the kernels of one leg (tools/gap_leg.sh)
#         intercept                      fixed cost per launch of the headline kernels, batch 128 vs 256 (tools/intercept.sh)
#         pgemmpmc[:M:N:K]               SQ counters of pgemm_nt and its compiled-out variants (tools/pgemm_counters.sh)
tag=$1; shift
mkdir -p gpurun_out
for task in "$@"; do
  IFS=: read -r name a1 a2 a3 a4 <<< "$task"
  out=gpurun_out/${tag}_${name}${a1:+_$(echo "$a1" | tr -c 'A-Za-z0-9\n' '_' | cut -c1-24)}.txt
  echo "=== $task -> $out"
  case $name in
    tests)   (timeout 2700 python -m pytest tests -m gpu -q ${a1:+-k "$a1"} 2>&1 | grep -v amdgpu.ids | tail -15) > $out ;;
    fuzz)    (timeout 900 python tools/fuzz_paths.py ${a1:-24} ${a2:-11} 2>&1 | grep -E "^(ok|FAIL|worst)" | cut -c1-60,250-420) > $out ;;
    smoke)   (timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3) > $out ;;
    bench)   (timeout 900 python bench.py $a1 2>&1 | grep -v amdgpu.ids | tail -3) > $out ;;
    harness) (timeout 600 tools/$a1 $a2 $a3 2>&1 | tail -200) > gpurun_out/${tag}_${a1}.txt; out=gpurun_out/${tag}_${a1}.txt ;;
    ab)      (timeout 1200 bash tools/ab_env.sh $a1 $a2 $a3 ${a4:-3} 2>&1) > $out ;;
    legs)    (timeout 2400 bash tools/ab_legs.sh ${a1:-2} 2>&1) > $out ;;
    profiles) (bash tools/collect_profiles.sh $tag 2>&1 | tail -30) > $out ;;
    sq)      (bash tools/collect_sq_counters.sh $tag 2>&1 | tail -30) > $out ;;
    py)      (timeout 1800 python tools/$a1 $a2 $a3 $a4 2>&1 | grep -v amdgpu.ids | tail -100) > gpurun_out/${tag}_${a1%.py}.txt; out=gpurun_out/${tag}_${a1%.py}.txt ;;
    legab)   (timeout 2400 bash tools/env_ab.sh $a1 $a2 ${a3:-2} $(echo "$a4" | tr ',' ' ') 2>&1) > $out ;;
    legprof) (bash tools/prof_leg.sh ${tag}_$a1 $a1 2>&1 | cut -c1-200) > $out ;;
    gaps)    (bash tools/gap_leg.sh ${tag}_gap $a1 2>&1 | cut -c1-200) > $out ;;
    intercept) (bash tools/intercept.sh ${tag}_ic 2>&1 | cut -c1-160) > $out ;;
    pgemmpmc) (bash tools/pgemm_counters.sh ${tag}_pgemm $a1 $a2 $a3 2>&1 | cut -c1-330) > $out ;;
    *)       echo "unknown task $task" ;;
  esac
  tail -40 $out | cut -c1-400
done
