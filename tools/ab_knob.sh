# same-box A/B of a Python-side knob: tools/ab_knob.sh KNOB "<bench args>"   (runs KNOB=1 (off) and KNOB unset, twice)
K=$1; shift
for r in 1 2; do for v in 1 0; do
if [ $v = 1 ]; then export SRK_DEBUG=1; export $K=1; else unset $K; fi
python bench.py "$@" --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$K=$v', d['config'].get('workload', d['config']), d['value'], d['ms_per_step'])"
done; done
