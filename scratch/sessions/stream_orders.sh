for o in "" "Sm,Tm,Sw,Tw,Sa" "Sm,Tm,Tw,Sw,Sa" "Sm,Tm,Sa,Sw,Tw" "Sm,Tm,Sw,Sa,Tw" "Sm,Tm,Tw,Sa,Sw" "Sm,Sw,Tm,Tw,Sa" "Sm,Tm,Sw,Tw,Saw,Sa" "Tm,Sm,Sw,Tw,Sa" "Sm,Tm,Sa,Tw,Sw"; do
  timeout 200 python scratch/probe_stream_order.py $o 2>&1 | grep -v amdgpu | tail -1
done
