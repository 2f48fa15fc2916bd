export LPIPM_EXPERIMENTAL=1 LPIPM_HALVES=0
for S in 128 256 512 1024; do
  echo -n "super=$S: "; LPIPM_SUPER=$S python scripts/adat_variants.py c4only 2>&1 | grep lockstep
done
