cd /root/repo
AB_CONFIG=c3 bash profiles/scripts/gpu_ab.sh "RG_QR_STREAMS=1" "RG_QR_STREAMS=0"
