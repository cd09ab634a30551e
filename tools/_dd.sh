cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu -k "ddbpn or dbpn or slice" 2>&1 | tail -3
for i in 1 2; do timeout 600 python bench.py --model ddbpn --batch 16 --steps 30 --no-cpu-baseline --no-other-configs --no-roofline --sustain-seconds 0 2>/dev/null | tail -1 | cut -c80-200; done
