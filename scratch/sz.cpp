#include "../mapper_amd/csrc/xm_wave.h"
#include <cstdio>
using namespace xm;
int main(){ printf("LightSE %zu LightPE %zu Heavy %zu  WMate(SE) %zu WAligner(SE) %zu\n", sizeof(WaveLdsT<WCfgLightSE>), sizeof(WaveLdsT<WCfgLightPE>), sizeof(WaveLdsT<WCfgHeavy>), sizeof(WMateT<WCfgLightSE>), sizeof(WAlignerT<WCfgLightSE>)); }
