#!/usr/bin/env bash
# scratch/ab_other.sh <model> "ENV=.. ENV=.." "-" ...: same-box A/B of one of the other configs (other_one.py), two repetitions
m=$1; shift
for rep in 1 2; do
  for cfg in "$@"; do
    [ "$cfg" = "-" ] && envs="" || envs="$cfg"
    printf 'rep %d [%-50s] ' $rep "$cfg"; env $envs timeout 300 python scratch/other_one.py $m 30 2>&1 | grep "ms per iteration"
  done
done
