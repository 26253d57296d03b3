R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04g; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/tests.log
python tools/train_profile.py lpdnet bf16 > $O/train_profile_bf16.txt 2>&1
python tools/train_profile.py lpdnet f32 > $O/train_profile_f32.txt 2>&1
python bench.py --no-cpu-baseline --no-secondary > $O/bench_line.json 2> $O/bench.err
tail -6 $O/tests.log; head -3 $O/train_profile_bf16.txt; head -3 $O/train_profile_f32.txt; python - <<'PY'
import json
r=json.load(open('gpurun_out/r04g/bench_line.json'))
print(r['value'], r['ms_per_step'], r['train']['ms_per_step'], r['train_bf16']['ms_per_step'], r['train']['peak_hbm_gib'], r['train_bf16']['peak_hbm_gib'])
PY
