cd /root/repo
bash profiles/scripts/gpu_ab.sh "RG_SORT_INDICES=0" "RG_SORT_INDICES=1"
AB_CONFIG=c4 bash profiles/scripts/gpu_ab.sh "RG_SORT_INDICES=0" "RG_SORT_INDICES=1" | cut -c1-120
