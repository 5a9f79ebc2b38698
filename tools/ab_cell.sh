for shape in "128 250 512" "256 250 1024" "512 250 1024" "1024 100 1024" "64 250 128"; do
  for L in cellold main; do
    if [ "$L" = main ]; then F=sparch_amd/libsparch_hip.so; else F=sparch_amd/libsparch_hip_$L.so; fi
    for kind in adLIF LIF; do
      echo -n "$shape $kind $L: "; SPARCH_HIP_LIB=$F timeout -k 10 100 python tools/rec_time.py $kind $shape 2>&1 | tail -1
    done
  done
done
