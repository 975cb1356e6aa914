#!/bin/bash
# rocm-smi beside the MFMA loop (tools/probe/mfma_clock_probe.hip), both shapes, twice.  Usage (GPU box): bash tools/probe/mfma_clock_probe.sh
cd "$(dirname "$0")"
for shape in 16 32 16 32; do
  timeout -k 5 30 ./mfma_clock_probe $shape 5 & pid=$!
  sleep 1.5
  for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket|Power" | tr '\n' ' '; echo; sleep 0.5; done
  wait $pid || exit 1
done
