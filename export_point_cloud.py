#!/usr/bin/env python3
"""export_point_cloud.py -- same command line as the reference's export_point_cloud.py (acezero_amd/cli.py)."""
import sys

from acezero_amd.cli import export_point_cloud_main

if __name__ == "__main__":
    sys.exit(export_point_cloud_main())
