for m in none tickets sites gen empty_cache ticketzero; do echo "=== $m"; python tools/_repro.py $m 2>&1 | grep -v Warn | grep -E "^H|^three|^state|^two|nonzero"; done
