mkdir -p gpurun_out/s2x
DLWP_BENCH_BACKEND=gloo DLWP_BENCH_ONE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 --no-other-configs --no-cpu-baseline > gpurun_out/s2x/bench2.json 2> gpurun_out/s2x/bench2.err
tail -2 gpurun_out/s2x/bench2.err
python -c "
import json
d=json.loads(open('gpurun_out/s2x/bench2.json').read().strip().split('\n')[-1])   # the LAST stdout line is the contract line (gloo prints above it)
print(d['n_gpus'], d['value'], d['ms_per_step'], d['scaling'], d['config']['parallelism'][:80])
"
DLWP_BENCH_BACKEND=gloo DLWP_BENCH_ONE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 6 --warmup 2 --no-other-configs --no-cpu-baseline --collect gather > gpurun_out/s2x/bench2g.json 2> gpurun_out/s2x/bench2g.err
tail -2 gpurun_out/s2x/bench2g.err
python -c "
import json
d=json.loads(open('gpurun_out/s2x/bench2g.json').read().strip().split('\n')[-1])
print(d['n_gpus'], d['value'], d['ms_per_step'], d['config']['collect'])
"
