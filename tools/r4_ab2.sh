for o in "den_split=1" "den_split=0" "den_split=1" "den_split=0"; do
python3 bench.py --no-alt --no-parity --no-cpu-baseline --no-also --option $o 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 $o', j['ms_per_step'], j['roofline']['frac'], [(h['kernel'], h['ms_per_step']) for h in j['roofline_hbm']])"
done
