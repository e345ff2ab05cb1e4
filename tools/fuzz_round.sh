# GPU fuzz campaign of a round: the four generators, every set-up in both launch modes, against the oracle.
#   fuzz_round.sh <tag> <first seed> <seeds per generator>
R=$GRAFT_REPO_ROOT; TAG=$1; FIRST=$2; N=$3; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
: > $O/fuzz_campaigns.txt
k=0
for g in v1 v2 v3 v4; do
  python3 tools/fuzz_campaign.py $((FIRST + k * 10000)) $N $g 2>&1 | tail -2 | tee -a $O/fuzz_campaigns.txt
  k=$((k + 1))
done
