# round 4, first GPU call: full -m gpu suite, cfg2 figures, train profiles (before), default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/tests.log
LPD_TEST_VERBOSE=1 python -m pytest tests/test_train_gpu.py -m gpu -q -s -k cfg2_full 2>&1 | grep -a "cfg2\|passed\|failed" > $O/cfg2.log
python tools/train_profile.py lpdnet bf16 > $O/train_profile_bf16_before.txt 2>&1
python tools/train_profile.py lpdnet f32 > $O/train_profile_f32_before.txt 2>&1
( time python bench.py ) > $O/bench_line.json 2> $O/bench.err
tail -3 $O/tests.log; cat $O/cfg2.log; head -3 $O/train_profile_bf16_before.txt; tail -4 $O/bench.err; head -c 600 $O/bench_line.json
