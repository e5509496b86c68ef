// conv.cpp -- FirDirect / Conv / Resample stages (placeholder until the FFT engine lands)
#include "stages.h"
namespace dspamd {
Stage *make_conv_stage(const Spec &sp, int, ssize_t, CascadeStage *)
{
	set_error("%s: error: this stage type is not implemented yet", sp.name.c_str());
	return nullptr;
}
}
