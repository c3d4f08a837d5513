cd tools
for i in 1 2 3; do for v in old cur; do for g in "256 1024" "768 512"; do ./kb_$v 10000000 21 $g 20 $v 16 | cut -c1-120; done; done; done
