for sz in 64M 256M 1G; do for it in 3 10; do for rep in 1 2; do
  timeout -k 5 60 build/bin/gpumon-diag pcie --bytes $sz --iters $it --no-expect --json | python3 -c "
import json,sys; p=json.load(sys.stdin)['pcie']; print('$sz', $it, round(p['h2d_gbps'],1), round(p['d2h_gbps'],1), round(p['bidir_h2d_gbps'],1), round(p['bidir_d2h_gbps'],1), round(p['kernel_read_gbps'],1), round(p['kernel_write_gbps'],1))" || exit 1
done; done; done
