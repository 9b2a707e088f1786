#!/usr/bin/env python3
"""register_mapping.py -- same command line as the reference's register_mapping.py, MI355X DSAC* (acezero_amd/cli.py)."""
import sys

from acezero_amd.cli import register_main

if __name__ == "__main__":
    sys.exit(register_main())
