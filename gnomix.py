#!/usr/bin/env python3
"""Same command line as the reference's gnomix.py (pre-trained mode, gnomix.py:318-355):

    python3 gnomix.py <query_file> <output_basename> <chr_nr> <phase> <path_to_model>

served by the MI355X path (gnomix_amd.cli)."""
import sys

from gnomix_amd.cli import main

if __name__ == "__main__":
    sys.exit(main(sys.argv))
