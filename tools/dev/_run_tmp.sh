mkdir -p gpurun_out/batchw; rm -f gpurun_out/batchw/log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batched_parameter_work or step_wgrads or backward_against_autograd_oracle or reduce_the_loss or data_parallel or captured or tape or cache" > gpurun_out/batchw/t.log 2>&1
tail -3 gpurun_out/batchw/t.log
for m in cifar10 mnist smap; do
  timeout -k 10 300 python tools/train_graph_bench.py $m 256 50 2>&1 | tail -1 >> gpurun_out/batchw/log
done
cut -c1-100 gpurun_out/batchw/log
