#!/usr/bin/env python3
"""Same command line as the reference's gnomix.py (pre-trained mode, gnomix.py:318-355):

    python3 gnomix.py <query_file> <output_basename> <chr_nr> <phase> <path_to_model>

served by the MI355X path (gnomix_amd.cli)."""
import os
import sys

from gnomix_amd.cli import main

if __name__ == "__main__":
    rc = main(sys.argv)
    # the outputs are written and closed: leave without tearing down the GPU runtime and its page-locked buffers (~0.2 s)
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(int(rc or 0))
