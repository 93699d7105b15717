cd scripts/probe
for fp in 2 512; do for pat in 0 1 2 3 4; do ./dma_rate $pat 256 4 512 8 $fp; done; done
for w in 1 2 4 16; do ./dma_rate 0 256 4 512 $w 2; done
./dma_rate 0 256 8 256 8 2; ./dma_rate 1 256 8 256 8 2; ./dma_rate 0 512 4 256 8 2; ./dma_rate 0 256 1 512 8 2; ./dma_rate 0 256 2 512 8 2
