# stage stamps of the exact title mix's refine launch (experiments build, DAE_DBG_MR): cycles of row 0's workgroup per stage
cd $GRAFT_REPO_ROOT
DAE_DBG_MR=1 DAE_LIB_AB=$GRAFT_REPO_ROOT/scripts/probe/libdae_hip_exp.so timeout 300 python scripts/time_title.py exact_bf16 20 2>&1 | grep "MIX_REFINE\|playlists/s" | tail -8
