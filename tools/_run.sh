# scratch: the command list of the builder's last gpurun call (bash tools/_run.sh on the GPU box); see tools/profile_all.sh,
# tools/profile_fast.sh, tools/ab_libs.sh, tools/gaps.py for the measurement recipes that are kept.
