"""python mimgnet/main.py --data-dir D --save-dir S [flags]: the reference's mimgnet/main.py command line
(flags and defaults in cmvae_amd/main.py) on the MI355X hot path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cmvae_amd.main import main  # noqa: E402

if __name__ == '__main__':
    main(['mimgnet'] + sys.argv[1:])
