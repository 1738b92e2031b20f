echo "# shader clock inside k_down2_mfma (random operands, 256 frames), tools/lab/down2_lab.hip -DD2_TRACE:"
echo "# every workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) at its start and end"
for l in E1 E2 E3; do for i in 1 2; do tools/lab/bin/d2lab_trace $l 256 | grep -E "median|shader clock"; done; done
echo "# the same silicon with CONSTANT operands (tools/lab/issue_probe.hip reaches 155.7 TFLOP/s = 2.4 GHz x 64 FLOP/clk x 1024 SIMDs)"
