export HOST_PROFILE=0
for w in c4_gdino_dec c2_q1k c2_q10k:Q=2000 c2_q10k:Q=3000 c2_q5k c4_gdino_dec:B=4 c4_gdino_dec:B=16; do
  for o in 0 1; do
    echo "== $w overlap=$o"; python tools/host_overhead.py $w 1000 overlap=$o 2>&1 | grep "us/step"
  done
done
