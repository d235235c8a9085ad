export TMPDIR=/tmp
for w in 3 4 5 6 7; do
  rm -rf gpurun_out/ph$w
  MLLM_HIP_HEAD_WPC=$w timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/ph$w -o ph --output-format csv -- python3 bench.py --steps 32 --warmup 4 > /dev/null 2>&1
  echo "wpc $w: $(grep dec_head gpurun_out/ph$w/*stats.csv | awk -F, '{print $(NF-6)}')"
done
