#!/bin/bash
# the other BASELINE configurations on one box (stage timers): cfg5 forward-only, cfg2, cfg3c, with the pair sort beside the row-segment sort
root=${GRAFT_REPO_ROOT:-$(pwd)}; cd $root
run () { timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-model-step "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stage_ms']; print('[$*]', d['value'], 'R', d['config']['R'], {k: round(v, 4) for k, v in s.items()})"; }
run --config cfg5 --forward-only
run --config cfg5 --forward-only --set tile_sort_rows=0
run --config cfg2
run --config cfg2 --set tile_sort_rows=0
run --config cfg3c
run --config cfg3c --set tile_sort_rows=0
run --config cfg4
