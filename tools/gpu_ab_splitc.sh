OUT=gpurun_out/r3j; mkdir -p $OUT
for arch in hrnet res_50 dla_34; do for sc in 1 0; do
  B=8; [ $arch = dla_34 ] && B=16
  CP_WINO_SPLITC=$sc timeout 300 python bench.py --arch $arch --batch $B --steps 60 --warmup 15 --no-cpu-baseline --no-profile > $OUT/b_${arch}_$sc.json 2>/dev/null
  python -c "
import json; l=json.load(open('$OUT/b_${arch}_$sc.json')); print('$arch B=$B splitc=$sc', l['value'], l['ms_per_step'])"
done; done
for arch in hrnet res_50; do for sc in 1 0; do
  CP_WINO_SPLITC=$sc timeout 300 python bench.py --arch $arch --batch 1 --steps 60 --warmup 15 --no-cpu-baseline --no-profile > $OUT/b1_${arch}_$sc.json 2>/dev/null
  python -c "
import json; l=json.load(open('$OUT/b1_${arch}_$sc.json')); print('$arch B=1 splitc=$sc', l['value'], l['ms_per_step'])"
done; done
