for i in 1 2 3 4 5 6; do python tools/probe_build_repeat.py 2 2>&1 | tail -2 | tr '\n' ' '; echo; done
