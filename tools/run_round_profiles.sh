#!/bin/bash
# The round's ONE profile regeneration (round-5 review: once, at the end): GPU test tier, smoke, the default workload's bench / stats / trace / PMC passes, the like-for-like
# traffic row, LeNet and AllConvNet profiles, the no-flags bench.   gpurun --timeout 4000 -- "bash tools/run_round_profiles.sh"
mkdir -p gpurun_out/r06f
python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r06f/gputests.log; cat gpurun_out/r06f/gputests.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/run_profiles.sh gpurun_out/r06f/vgg > gpurun_out/r06f/run_profiles.log 2>&1; tail -5 gpurun_out/r06f/run_profiles.log
VARIANTS=base bash tools/conv_traffic_ablation.sh gpurun_out/r06f/abl > gpurun_out/r06f/conv_traffic_ablation.txt 2>&1; cat gpurun_out/r06f/conv_traffic_ablation.txt | head -8
bash tools/run_profiles_wl.sh gpurun_out/r06f/lenet lenet > gpurun_out/r06f/lenet.log 2>&1; tail -3 gpurun_out/r06f/lenet.log
bash tools/run_profiles_wl.sh gpurun_out/r06f/allconv allconv > gpurun_out/r06f/allconv.log 2>&1; tail -3 gpurun_out/r06f/allconv.log
python3 bench.py > gpurun_out/r06f/bench_noflags.json 2> gpurun_out/r06f/bench_noflags.log; cat gpurun_out/r06f/bench_noflags.json | cut -c1-1200
