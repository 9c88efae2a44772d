for c in 1 4 16 40 100; do echo "copies $c:"; bash tools/time_variants.sh $c auto paired48 paired24 v6l256 v6l128 v5s512; done
