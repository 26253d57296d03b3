# round-4 profile set (on the GPU box): bash tools/prof_r04.sh   -> gpurun_out/r04prof (summaries copied to profiles/ by hand)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04prof; mkdir -p $O; cd $R
python3 tools/kagg_pmc.py cfg2 cfg5 > $O/kagg_pmc.log 2>&1
cp profiles/kagg_pmc.json $O/kagg_pmc.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r04 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/stats.err
cd $R
find $O -name "*kernel_stats*" -exec cp {} $O/ \;
find $O -type f -size +3M -delete
python3 tools/train_profile.py lpdnet bf16 > $O/train_profile_bf16.txt 2>&1
python3 tools/train_profile.py lpdnet f32 > $O/train_profile_f32.txt 2>&1
python3 bench.py > $O/bench_line.json 2> $O/bench.err
tail -5 $O/kagg_pmc.log; ls $O | head -20; head -c 400 $O/bench_line.json
