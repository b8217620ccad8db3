#!/bin/bash
# After ANY edit under csrc/ or include/: regenerate the two tracked static-evidence files (no GPU, ~3 min).
#   profiles/r06_isa_diff_since_298c878.txt  every kernel vs the last GPU-run tree (tests/test_host_logic.py checks its digest)
#   profiles/r06_variants_isa.txt            default vs opt-in variant: registers, loop mix, issue model
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
bash $R/tools/isa_diff_since.sh 298c878 > $R/profiles/r06_isa_diff_since_298c878.txt
bash $R/tools/variants_isa.sh > $R/profiles/r06_variants_isa.txt
grep -c '^same' $R/profiles/r06_isa_diff_since_298c878.txt
grep -E '^DIFFERS|new file' $R/profiles/r06_isa_diff_since_298c878.txt | cut -c1-90
