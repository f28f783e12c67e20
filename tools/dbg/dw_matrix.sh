# depthwise experiments: channel octets per block x XCD-aware order
for ob in 16 36 72; do for x in 0 1; do
  echo "OB=$ob XCD=$x"
  PTOCR_DW_OB=$ob PTOCR_DW_XCD=$x bash tools/dbg/bf16_trace.sh | grep "dwconv\|sum" | awk '{printf "%s ", $(NF-1)} END {print ""}'
done; done
