#!/bin/bash
# Samples the shader clock while a command runs:  bash tools/clock_probe.sh <cmd...>   (diagnostic: MFMA-heavy kernels lower the clock)
"$@" > /tmp/clock_probe_cmd.log 2>&1 &
PID=$!
sleep 4
for i in $(seq 1 12); do
  rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1
  rocm-smi --showpower 2>/dev/null | grep -i "power" | head -1
  sleep 0.4
done
wait $PID
tail -2 /tmp/clock_probe_cmd.log
