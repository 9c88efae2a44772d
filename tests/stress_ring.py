import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import rust_lz_fear_amd
from rust_lz_fear_amd import ffi
ffi.lib().lzf_frame_set_pinned_limit(16 << 20)      # two 4 MiB slots: every call of more than 8 MiB goes round the ring
import stress_frames_many, stress_frames
for seed in (701, 702, 703):
    sys.argv = ["x", "25", str(seed)]; stress_frames_many.main()
sys.argv = ["x", "30", "704"]; stress_frames.main()
print(ffi.frame_stats())
