# serial (one-stream) kernel durations of the eval step: bash tools/prof_eval_serial.sh   (GPU box; output gpurun_out/evalserial)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evalserial
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
LPD_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s0 -o e -- python3 $R/bench.py --steps 10 --warmup 3 --no-train --no-cpu-baseline --no-secondary > $O/line_s0.json 2> $O/s0.err
find $O -name "*kernel_stats*" -exec cp {} $O/serial_kernel_stats.csv \;
find $O -type f -size +3M -delete
cd $R
python3 bench.py --no-train --no-cpu-baseline --no-secondary > $O/line.json 2> $O/line.err
LPD_SIDE_STREAM=0 python3 bench.py --no-train --no-cpu-baseline --no-secondary > $O/line_serial.json 2>> $O/line.err
cat $O/line.json $O/line_serial.json
