#!/usr/bin/env python3
"""python scripts/run_with_watchdog.py SECONDS script.py [args...]: run a script; after SECONDS dump every thread's Python stack to stderr and exit
(where is a hung benchmark waiting?)."""
import faulthandler, runpy, sys
secs = float(sys.argv[1])
faulthandler.enable()
faulthandler.dump_traceback_later(secs, exit=True)
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
