"""`python inference.py -c conf/...yaml -m ckpt.pth --input_dir ... --output_dir ...` - the reference's CLI
(same flags), served by the MI355X engine.  See srgd_amd/inference.py."""
from srgd_amd.inference import main

if __name__ == "__main__":
    main()
