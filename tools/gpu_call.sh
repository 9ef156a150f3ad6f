#!/bin/bash
# One gpurun call:  gpurun -- 'bash tools/gpu_call.sh <tag> <step> [<step> ...]'
# Output goes to gpurun_out/<tag>/; a step that times out ends the call (no further GPU step is started after a kill).
# Steps:  env:VAR=VALUE | unset:VAR | tests | tests-noexit | tests:<pytest -k expression> | smoke | bench | bench:<bench.py args> | configs | scan | panel | prof | pmc | py:<script and args>
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=gpurun_out/$tag
mkdir -p $R/$out
export TMPDIR=/tmp
cd $R
step() { # name, seconds, command...
  local name=$1 secs=$2; shift 2
  local t0=$(date +%s)
  timeout -k 10 $secs "$@"; local rc=$?
  echo "$name rc=$rc ($(( $(date +%s) - t0 )) s)" | tee -a $R/$out/summary.txt
  if [ $rc -ge 124 ]; then echo "stopping after $name" | tee -a $R/$out/summary.txt; exit 1; fi
}
for s in "$@"; do
  case $s in
    tests) step "all gpu tests" 1000 bash -c "python -m pytest tests -q -m gpu -x > $out/tests_all.log 2>&1" ;;
    tests-noexit) step "all gpu tests (no -x)" 1100 bash -c "python -m pytest tests -q -m gpu > $out/tests_all.log 2>&1" ;;
    tests:*) step "gpu tests -k ${s#tests:}" 900 bash -c "python -m pytest tests -q -m gpu -x -k '${s#tests:}' > $out/tests_sel.log 2>&1" ;;
    smoke) step "smoke" 200 bash -c "python -c 'import __graft_entry__ as g; g.smoke()' > $out/smoke.log 2>&1" ;;
    bench) n=bench; [ -e $out/bench.json ] && n=bench_$(date +%s | tail -c 4)
      step "bench ($n)" 300 bash -c "python bench.py > $out/$n.json 2> $out/$n.err" ;;
    bench:*) n=$(echo "${s#bench:}" | tr -c 'A-Za-z0-9_.\n' '_' | cut -c1-40)_$(date +%s | tail -c 4)
      step "bench ${s#bench:}" 300 bash -c "python bench.py --no-cpu-baseline ${s#bench:} > $out/bench_$n.json 2> $out/bench_$n.err" ;;
    configs)
      for cfg in "c3 --points 2048 --profile final" "c3_all --points 2048 --profile all" "c4rank --batch 8 --points 4096" "b32n4096 --points 4096" \
                 "x3 --precision bf16x3" "f32act --precision bf16_f32act" "final1024 --profile final" "all1024 --profile all"; do
        set -- $cfg; t=$1; shift
        step "bench $t" 200 bash -c "python bench.py --steps 100 --warmup 10 --no-cpu-baseline $* > $out/bench_$t.json 2> $out/bench_$t.err"
      done ;;
    scan) n=scan; [ -e $out/scan.json ] && n=scan_$(date +%s | tail -c 4)
      step "scan pipeline ($n)" 200 bash -c "python tools/bench_scan.py > $out/$n.json 2> $out/$n.err" ;;
    panel) step "panel probe" 200 bash -c "python tools/panel_probe.py > $out/panel_probe.jsonl 2> $out/panel_probe.err" ;;
    prof)
      cd /tmp
      step "prof c2" 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o c2 -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline
      step "prof c3" 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o c3 -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --points 2048 --profile final
      step "prof b32n4096" 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o b32n4096 -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --points 4096
      cd $R; rm -f $out/prof/*kernel_trace* ;;
    pmc)
      cd /tmp
      step "pmc fetch" 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$out/pmc_fetch -o f -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-graph
      step "pmc write" 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$out/pmc_write -o w -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-graph
      cd $R
      step "pmc summary" 60 python tools/pmc_summary.py $out/pmc_fetch $out/pmc_write $out/pmc r3
      rm -f $out/pmc_fetch/*kernel_trace* $out/pmc_write/*kernel_trace* ;;
    rprof:*) a="${s#rprof:}"; n="${a%%:*}"; sc="${a#*:}"
      cd /tmp
      step "rprof $n" 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/rprof -o $n -- python3 $R/$sc
      cd $R; case "$n" in trace*) ;; *) rm -f $out/rprof/*kernel_trace* ;; esac ;;
    env:*) export "${s#env:}"; echo "env ${s#env:}" | tee -a $R/$out/summary.txt ;;
    unset:*) unset "${s#unset:}" ;;
    py:*) n=$(echo "${s#py:}" | tr -c 'A-Za-z0-9_.\n' '_' | cut -c1-40)
      n="${n}_$(date +%s | tail -c 4)"
      step "py ${s#py:}" 400 bash -c "python ${s#py:} > $out/$n.out 2> $out/$n.err" ;;
    *) echo "unknown step $s" | tee -a $R/$out/summary.txt ;;
  esac
done
