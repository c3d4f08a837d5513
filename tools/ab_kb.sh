cd tools
for i in 1 2 3; do for v in old cur; do ./kb_$v 10000000 21 768 512 20 $v 16 | cut -c1-175; done; done
