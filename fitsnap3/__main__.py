import sys

from fitsnap_amd.cli import main

if __name__ == "__main__":
    sys.exit(main())
