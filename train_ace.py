#!/usr/bin/env python3
"""train_ace.py -- same command line as the reference's train_ace.py, MI355X head trainer (acezero_amd/cli.py)."""
import sys

from acezero_amd.cli import train_main

if __name__ == "__main__":
    sys.exit(train_main())
