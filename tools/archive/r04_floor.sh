# what bounds the set-up when the factorisation chain is taken away (timing aid TP_CD_STAGES: 1 = fill only, 2 = + factor)
cd /tmp
for st in 3 2 1; do echo "TP_CD_STAGES=$st"; TP_CD_STAGES=$st python $GRAFT_REPO_ROOT/tools/r04_setup.py 2>&1 | grep set-up; done
