cd scratch/xtb
for b in x3base x3nomult x3nofetch; do echo -n "$b: "; ./xtb_$b; done
