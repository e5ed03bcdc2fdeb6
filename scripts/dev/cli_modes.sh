set -e
cd $GRAFT_REPO_ROOT
D=/dev/shm/helen_cli_modes; rm -rf $D; mkdir -p $D
python - <<'PY'
import sys
from helen_amd.synthetic import write_image_dir
from helen_amd.model_handler import ModelHandler
from helen_amd.weights import make_weights
write_image_dir("/dev/shm/helen_cli_modes/img", 300, n_files=3, seed=77, short_every=9)
ModelHandler.save_model(make_weights(seed=11, input_scale=1.0/64.0), None, 128, 1, 0, "/dev/shm/helen_cli_modes/m.pkl")
PY
for p in fp32 fp32x3 bf16; do
  HELEN_PRECISION=$p timeout 300 python -m helen_amd polish -i $D/img -m $D/m.pkl -b 64 -w 4 -o $D/out_$p -p asm -g -d_ids 0 > $D/log_$p.txt 2>&1 || { tail -5 $D/log_$p.txt; exit 1; }
  ls $D/out_$p/predictions_*/ | tr '\n' ' '; grep -c ">" $D/out_$p/asm.fa; md5sum $D/out_$p/asm.fa | cut -c1-12
done
cmp $D/out_fp32/asm.fa $D/out_fp32x3/asm.fa && echo "fp32 == fp32x3 FASTA identical"
python - <<'PY'
a=open("/dev/shm/helen_cli_modes/out_fp32/asm.fa").read().split("\n")[1]
b=open("/dev/shm/helen_cli_modes/out_bf16/asm.fa").read().split("\n")[1]
print("len fp32", len(a), "len bf16", len(b), "identical" if a==b else "differs (bf16 arithmetic)")
PY
rm -rf $D
