cd $GRAFT_REPO_ROOT
AB_TILES=10 python tools/abbench.py > /dev/null 2>&1
for v in al6 abl_NO_DEPS abl_NO_LIT abl_NO_FARPUT abl_NO_NEAR abl_NO_FLUSH abl_PARSEONLY; do
  KPROF_SETS=1 AB_TILES=10 bash tools/kprof.sh r3c_$v libzxc_$v.so > /dev/null 2>&1
  echo "== $v"; grep -E "SQ_INSTS|bench line" gpurun_out/r3c_${v}_kprof.txt
done
