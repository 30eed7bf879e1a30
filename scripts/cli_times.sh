#!/bin/bash
# end-to-end wall time of the command line (process start to exit, output files on local disk)
P=$(pwd)/pansim_amd/pansim
t() { local name=$1; shift; local a=$(date +%s%N); "$@" > /dev/null 2>&1; local b=$(date +%s%N); echo "$name $(( (b - a) / 1000000 )) ms"; }
AUTH="--n_gen 100 --pop_size 1000 --core_size 1342000 --pan_genes 4400 --core_genes 1342 --avg_gene_freq 0.45 --threads 4 --core_mu 0.019 --HR_rate 0.0 --HGT_rate 0.0 --rate_genes1 1.0 --rate_genes2 1000 --prop_genes2 0.0 --prop_positive -0.1 --pos_lambda 100 --neg_lambda 100 --verbose"
t warmup $P --pop_size 1000 --core_size 1200000 --pan_genes 6000 --n_gen 100 --seed 0 --outpref /tmp/o1
t cfg2 $P --pop_size 1000 --core_size 1200000 --pan_genes 6000 --n_gen 100 --seed 0 --outpref /tmp/o1
t authors_competition_0 $P $AUTH --outpref /tmp/o2 --competition_strength 0.0
t authors_competition_100 $P $AUTH --outpref /tmp/o2 --competition_strength 100
t authors_competition_1e4 $P $AUTH --outpref /tmp/o2 --competition_strength 10000
t authors_competition_1e8 $P $AUTH --outpref /tmp/o2 --competition_strength 100000000
t cfg4_10_generations $P --pop_size 65536 --n_gen 10 --outpref /tmp/o4
rm -f /tmp/o1* /tmp/o2* /tmp/o4*
