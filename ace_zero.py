#!/usr/bin/env python3
"""ace_zero.py -- same command line as the reference's ace_zero.py; the whole reconstruction loop runs in ONE process on one
MI355X (acezero_amd/session.py) instead of one subprocess per mapping / registration round."""
import sys

from acezero_amd.cli import ace_zero_main

if __name__ == "__main__":
    sys.exit(ace_zero_main())
