for q in 1 16 33 48 70 100 128; do echo "## Q=$q"; bash scripts/ab.sh ab/lib_prev.so ab/lib_nqb8.so --queries $q --steps 100 2>/dev/null | sort | awk '{print}' ; done
