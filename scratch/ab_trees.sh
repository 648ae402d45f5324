#!/usr/bin/env bash
# same-box A/B of the eager host path: the tree at scratch/old_tree (a git worktree of an earlier commit, built) against this one
for rep in 1 2; do
  for m in sagan srgan; do
    printf 'rep %d old  ' $rep; (cd scratch/old_tree && timeout 300 python scratch/other_one.py $m 30 2>&1 | grep "ms per iteration")
    printf 'rep %d new  ' $rep; timeout 300 python scratch/other_one.py $m 30 2>&1 | grep "ms per iteration"
  done
done
