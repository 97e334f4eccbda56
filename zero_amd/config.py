# coding: utf-8
"""HParams defaults of the hot path -- the run.py config contract.

Key names, value types and defaults follow the reference's ``global_params``
(run.py:24-239) so that an existing ``param.json`` / ``--config`` dict /
``--parameters`` string of the reference loads unchanged.  Keys that only
drive out-of-scope subsystems (RNN cells, l0drop, EMA ...) are kept as inert
values for that reason.
"""

from zero_amd.utils.hparams import HParams

# (name, default) in the reference's declaration order (run.py:24-239).
_DEFAULTS = [
    ("shared_source_target_embedding", False),
    ("shared_target_softmax_embedding", True),
    ("decode_length", 50), ("beam_size", 4), ("decode_alpha", 0.6),
    ("enable_noise_beam_search", False), ("beam_search_temperature", 1.0),
    ("top_beams", 1), ("search_mode", "cache"),
    ("max_relative_position", 16),
    ("nstable", 4), ("lrdecay_start", 600000), ("lrdecay_end", 1200000),
    ("warmup_steps", 400), ("lrate_strategy", "gnmt+"), ("lrate_decay", 0.5),
    ("lrate_patience", 1), ("cosine_period", 5000), ("cosine_factor", 1),
    ("estop_patience", 100),
    ("initializer", "uniform"), ("initializer_gain", 0.08),
    ("hidden_size", 1000), ("embed_size", 620),
    ("dropout", 0.1), ("relu_dropout", 0.1), ("residual_dropout", 0.1),
    ("label_smooth", 0.1),
    ("model_name", "rnnsearch"), ("scope_name", "rnnsearch"),
    ("cell", "atr"), ("caencoder", True), ("layer_norm", False),
    ("use_deep_att", False), ("swap_memory", True),
    ("filter_size", 2048), ("attention_dropout", 0.1),
    ("num_encoder_layer", 6), ("num_decoder_layer", 6), ("num_heads", 8),
    ("aan_mask", True), ("use_ffn", False),
    ("max_len", 100), ("eval_max_len", 1000000),
    ("batch_size", 80), ("token_size", 3000), ("batch_or_token", "token"),
    ("eval_batch_size", 32), ("shuffle_batch", True),
    ("strategies", ["aan"]),
    ("process_num", 1), ("buffer_size", 100),
    ("input_queue_size", 100), ("output_queue_size", 100),
    ("src_vocab_file", ""), ("tgt_vocab_file", ""),
    ("src_train_file", ""), ("tgt_train_file", ""),
    ("src_dev_file", ""), ("tgt_dev_file", ""),
    ("src_test_file", ""), ("tgt_test_file", ""),
    ("output_dir", ""), ("test_output", ""), ("pretrained_model", ""),
    ("beta1", 0.9), ("beta2", 0.999), ("epsilon", 1e-9),
    ("clip_grad_norm", 5.0), ("gnorm_upper_bound", 1e20),
    ("lrate", 1e-5), ("min_lrate", 0.0), ("max_lrate", 1.0),
    ("epoches", 10), ("update_cycle", 1), ("gpus", [0]),
    ("safe_nan", False), ("dl4mt_redict", True), ("ema_decay", -1.),
    ("data_leak_ratio", 0.5), ("deep_transformer_init", False),
    ("disp_freq", 100), ("eval_freq", 10000), ("save_freq", 5000),
    ("sample_freq", 1000), ("checkpoints", 5), ("best_checkpoints", 1),
    ("max_training_steps", 1000),
    ("nthreads", 6), ("random_seed", 1234), ("train_continue", True),
    ("default_dtype", "float32"), ("dtype_epsilon", 1e-8), ("dtype_inf", 1e8),
    ("loss_scale", 1.0),
    # build-specific (round 5): "bfloat16" = the product decode path; "float32" = fp32 masters / activations / accumulation
    # through zk_f32_* (zero_amd/models/_decode_f32.py): rounds where the reference's default dtype rounds
    ("decode_dtype", "bfloat16"),
    ("l0_norm_reg_scalar", 1.0), ("l0_norm_start_reg_ramp_up", 0),
    ("l0_norm_end_reg_ramp_up", 10000), ("l0_norm_warm_up", True),
]


def default_params():
    """A fresh HParams holding the reference defaults (run.py:24-239)."""
    return HParams(**dict(_DEFAULTS))


def transformer_base_params(**overrides):
    """Canonical Transformer-base recipe (docs/l0drop/README.md:81-115)."""
    p = default_params()
    p.override_from_dict(dict(
        hidden_size=512, embed_size=512, filter_size=2048, num_heads=8,
        num_encoder_layer=6, num_decoder_layer=6,
        dropout=0.1, relu_dropout=0.1, residual_dropout=0.1,
        attention_dropout=0.1, label_smooth=0.1,
        model_name="transformer", scope_name="transformer",
        initializer="uniform_unit_scaling", initializer_gain=1.0,
        lrate_strategy="noam", lrate=1.0, warmup_steps=4000,
        beta1=0.9, beta2=0.98, epsilon=1e-8, clip_grad_norm=0.0,
        token_size=6250, update_cycle=4, max_len=256,
    ))
    p.override_from_dict(overrides)
    return p


class SyntheticVocab(object):
    """Vocabulary of a given size with the reference's reserved ids
    (vocab.py:16-22: pad=0, unk=1, eos=2)."""

    def __init__(self, size):
        self._size = int(size)

    def size(self):
        return self._size

    def pad(self):
        return 0

    def unk(self):
        return 1

    def eos(self):
        return 2
