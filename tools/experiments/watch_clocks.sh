# shader clock / power of GPU 0 while a command runs: rocm-smi polled twice a second beside it (same GPU, separate process that only reads sysfs)
"$@" > gpurun_out/watch_cmd.log 2>&1 &
PID=$!
while kill -0 $PID 2>/dev/null; do
  rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | tr '\n' ' ' | cut -c1-300
  echo
  sleep 0.5
done
wait $PID
tail -c 300 gpurun_out/watch_cmd.log
