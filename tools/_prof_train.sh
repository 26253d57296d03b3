R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O; cd $R
python tools/train_profile.py lpdnet f32 > $O/train_profile_f32.txt 2>&1
python tools/train_profile.py lpdnet bf16 > $O/train_profile_bf16.txt 2>&1
python tools/tn_bench.py > $O/tn_bench.txt 2>&1
python bench.py > $O/bench_line.json 2> $O/bench.err
tail -c 3000 $O/bench_line.json
