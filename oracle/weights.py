"""Re-export of the synthetic data generators (they hold no model arithmetic and live in the
package so that bench.py's GPU leg does not import oracle/).  TEST INFRASTRUCTURE."""
from moonshine_amd.synth import (  # noqa: F401
    ARCHS,
    STREAMING_ARCHS,
    StreamingArchConfig,
    make_streaming_weights,
    streaming_tensor_specs,
    write_streaming_model_dir,
    ArchConfig,
    load_safetensors,
    make_audio,
    make_weights,
    save_safetensors,
    tensor_specs,
    write_model_dir,
)
