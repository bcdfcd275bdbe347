#!/bin/bash
# Shader clock and package power while a command runs (rocm-smi polled in the background):
#   tools/clock_power_sample.sh <tag> -- <cmd...>     -> gpurun_out/clk_<tag>.txt (one line per sample) + a summary on stdout
TAG=$1; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/clk_$TAG.txt; : > $O
( while true; do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | tr '\n' ' ' >> $O; echo >> $O; sleep 0.2; done ) &
SPID=$!
"$@"
kill $SPID 2>/dev/null
python3 - "$O" <<'PY'
import re, sys, statistics
clk, pw = [], []
for l in open(sys.argv[1]):
    m = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', l); p = re.search(r'Power \(W\): ([\d.]+)', l)
    if m: clk.append(int(m.group(1)))
    if p: pw.append(float(p.group(1)))
if clk: print(f"sclk MHz: n={len(clk)} min={min(clk)} median={statistics.median(clk)} max={max(clk)}")
if pw: print(f"power W: n={len(pw)} min={min(pw)} median={statistics.median(pw)} max={max(pw)}")
PY
