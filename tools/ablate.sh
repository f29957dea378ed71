#!/bin/bash
# Build ablation variants of conv.hip (sed-edited copies in /tmp) into neural-audio-fp_amd/_abl/.
# Used only for kernel-time breakdown experiments (guide: "ablate before optimizing").
set -e
R=/root/repo; S=$R/neural-audio-fp_amd/csrc; O=$R/neural-audio-fp_amd/_abl; mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc"
build() { # name, sed-expr...
  name=$1; shift; T=/tmp/abl_$name; rm -rf $T; mkdir -p $T/neural-audio-fp_amd/csrc $T/include
  cp $S/*.hip $S/*.h $T/neural-audio-fp_amd/csrc/; cp $R/include/nafp.h $T/include/
  for e in "$@"; do sed -i "$e" $T/neural-audio-fp_amd/csrc/conv.hip; done
  /opt/rocm/bin/hipcc $FLAGS $T/neural-audio-fp_amd/csrc/*.hip -o $O/libnafp_$name.so
  echo built $name
}
build base
build nostore 's/if (m < p.M) p.y\[/if (m < p.M \&\& v == 12345.678f) p.y[/'
build nogb 's/rg\[i\] = \*(const float4\*)(p.gamma_in + og);/rg[i] = make_float4(1.f,1.f,1.f,1.f);/; s/rb\[i\] = \*(const float4\*)(p.beta_in + og);/rb[i] = make_float4(0.f,0.f,0.f,0.f);/'
build noload 's/if (s + 1 < n_steps) load_step(s + 1);/if (s + 1 < n_steps \&\& p.M < 0) load_step(s + 1);/'
build nomfma 's/acc\[mi\]\[ni\] = __builtin_amdgcn_mfma_f32_32x32x2f32(a\[mi\]\.\([xyzw]\), b\[ni\]\.\([xyzw]\), acc\[mi\]\[ni\], 0, 0, 0);/acc[mi][ni][0] += a[mi].\1 * b[ni].\2;/'
build nostore_noload 's/if (m < p.M) p.y\[/if (m < p.M \&\& v == 12345.678f) p.y[/' 's/if (s + 1 < n_steps) load_step(s + 1);/if (s + 1 < n_steps \&\& p.M < 0) load_step(s + 1);/'
