#!/bin/bash
# Developer tool (GPU box): the bench's cfg2 loop alternating between this tree and another checked-out tree
# (e.g. `git worktree add build/r3tree <commit>` + its own build), same box, same flags.
# usage: tools/gpu_ab_trees.sh other_tree_dir [rounds] [extra bench flags]
other=$1; rounds=${2:-3}; shift; shift
for r in $(seq $rounds); do
  for tree in $other .; do
    (cd $tree && python bench.py --extra-configs none --no-cpu-baseline --total-photons 0 --repeats 7 --sustained-s 3 "$@" 2>/dev/null) | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tree', 'median %.4e sustained %.4e' % (d['value'], d['sustained']['value']))"
  done
done
