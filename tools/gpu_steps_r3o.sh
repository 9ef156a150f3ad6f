out=$1
R=$GRAFT_REPO_ROOT
step() { # name, seconds, command... ; a step that times out ends the call
  local name=$1 secs=$2; shift 2
  local t0=$(date +%s)
  timeout -k 10 $secs "$@"; local rc=$?
  echo "$name rc=$rc ($(( $(date +%s) - t0 )) s)" | tee -a $R/$out/summary.txt
  if [ $rc -ge 124 ]; then echo "stopping after $name" | tee -a $R/$out/summary.txt; exit 1; fi
}
step "probe prio0" 200 bash -c "PN_PANEL_PRIO=0 python tools/panel_probe.py > $out/probe_prio0.jsonl 2> $out/probe0.err"
step "probe prio1" 200 bash -c "PN_PANEL_PRIO=1 python tools/panel_probe.py > $out/probe_prio1.jsonl 2> $out/probe1.err"
step "probe prio0 again" 200 bash -c "PN_PANEL_PRIO=0 python tools/panel_probe.py > $out/probe_prio0b.jsonl 2> $out/probe0b.err"
step "stamps prio1" 200 bash -c "PN_PANEL_PRIO=1 python tools/panel_stamps.py > $out/stamps_prio1.jsonl 2> $out/stamps1.err"
