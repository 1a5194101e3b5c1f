#!/bin/bash
# LDS conflict counters of tools/ubench/lds_conflicts.hip's access patterns -> gpurun_out/pmc_lds_ubench.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
d=$R/gpurun_out/pmc_lds_ub
rm -rf $d
timeout -s KILL 60 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $d -o run -- $R/var/lds_conflicts > $d.log 2>&1
python3 - $(find $d -name '*.db' | head -1) > $R/gpurun_out/pmc_lds_ubench.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, avg(value) from counters_collection where kernel_name like '%k<%' group by kernel_name, counter_name").fetchall()
d = {}
for k, c, v in rows:
    d.setdefault(k, {})[c] = v
for k in sorted(d, key=lambda s: int(s.split('<')[1].split('>')[0].replace('(','').replace(')','').split()[-1]) if '<' in s else 0):
    v = d[k]
    print(f"{k[:28]:28s} insts {v.get('SQ_INSTS_LDS',0):10.0f} idx_active {v.get('SQ_LDS_IDX_ACTIVE',0):10.0f} bank_conflict {v.get('SQ_LDS_BANK_CONFLICT',0):10.0f}  cycles/inst {v.get('SQ_LDS_IDX_ACTIVE',0)/max(v.get('SQ_INSTS_LDS',1),1):5.2f} conflict/inst {v.get('SQ_LDS_BANK_CONFLICT',0)/max(v.get('SQ_INSTS_LDS',1),1):5.2f}")
PY
rm -rf $d
cat $R/gpurun_out/pmc_lds_ubench.txt
