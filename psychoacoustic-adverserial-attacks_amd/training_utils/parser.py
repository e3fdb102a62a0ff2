"""Command-line surface of the reference's ``src/training_utils/parser.py`` (same flags, types and
defaults — SURVEY §5.6) plus the non-breaking additions of SURVEY §8b."""
import argparse

NORM_CHOICES = ["l2", "linf", "snr", "tv", "fletcher_munson", "min_max_freqs", "max_phon"]


def _norm_type(v: str) -> str:
    parts = v.split("+")
    for part in parts:
        if part not in NORM_CHOICES:
            raise argparse.ArgumentTypeError(f"invalid choice: {v!r} (choose from {NORM_CHOICES}, or 'a+b')")
    return v


def create_arg_parser():
    parser = argparse.ArgumentParser()
    # standard training params (parser.py:10-20)
    parser.add_argument('--batch_size', type=int, default=64, help='batch size')
    parser.add_argument('--lr', type=float, default=1e-4, help='lr for the perturbation update')
    parser.add_argument('--early_stopping', type=int, default=4, help='how many epochs to wait before early stopping')
    parser.add_argument('--num_epochs', type=int, default=50, help='how many epochs at all')
    parser.add_argument('--optimizer_type', type=str, choices=["adam", "pgd"], default='adam',
                        help='how to optimize the perturbation update')
    parser.add_argument('--gamma', type=float, default=0.9, help='weight decay')
    parser.add_argument('--step_size', type=int, default=2, help='how many epochs does it take before we decay weights')
    parser.add_argument('--dataset', type=str, default="LibreeSpeech", choices=["LibreeSpeech", "CommonVoice", "tedlium"])
    parser.add_argument('--resume_from', type=str, default=None,
                        help='Path to a saved perturbation .pt file to resume training from')
    # adversarial params (parser.py:29-53)
    parser.add_argument('--target_reps', type=int, default=5)
    parser.add_argument('--target', type=str, default="delete", help='Target phrase for targeted attacks')
    parser.add_argument('--attack_mode', type=str, choices=["untargeted", "targeted"], default="untargeted")
    parser.add_argument('--norm_type', type=_norm_type, default='max_phon',
                        help='type of norm to limit the perturbation (extension: "a+b" applies a then b)')
    parser.add_argument('--fm_epsilon', type=float, default=2)
    parser.add_argument('--l2_size', type=float, default=0.05)
    parser.add_argument('--linf_size', type=float, default=0.0001)
    parser.add_argument('--snr_db', type=float, default=64)
    parser.add_argument('--min_freq_attack', type=float, default=120)
    parser.add_argument('--max_freq_attack', type=float, default=20_000)
    parser.add_argument('--tv_epsilon', type=float, default=0.001)
    parser.add_argument('--max_phon_level', type=float, default=20)
    # sound properties (parser.py:57-63)
    parser.add_argument('--phon_reference_db', type=float, default=65)
    parser.add_argument('--sr', type=int, default=16000)
    parser.add_argument('--n_fft', type=int, default=1024)
    parser.add_argument('--hop_length', type=int, default=256)
    parser.add_argument('--win_length', type=int, default=1024)
    parser.add_argument('--relative_audio_length', type=float, default=0.80)
    # others (parser.py:64-66)
    parser.add_argument('--seed', type=int, default=5)
    parser.add_argument('--small_data', action='store_true')
    parser.add_argument('--num_items_to_inspect', type=int, default=12)
    # additions: none changes the meaning of a reference flag
    parser.add_argument('--device', type=str, default="cuda")
    parser.add_argument('--model_path', type=str, default=None, help='local HF checkpoint directory (no hub access)')
    parser.add_argument('--arch', type=str, choices=["base", "large-lv60", "tiny"], default="base",
                        help='architecture for rule-generated weights when no --model_path is given')
    parser.add_argument('--dtype', type=str, choices=["bf16", "fp32"], default="bf16",
                        help='MFMA operand precision: bf16, or fp32 = split-bf16 (3 passes), fp32-parity')
    parser.add_argument('--audio_seconds', type=float, default=10.0, help='clip length for synthetic data')
    parser.add_argument('--steps_per_epoch', type=int, default=4, help='synthetic batches per epoch')
    parser.add_argument('--data_dir', type=str, default=None, help='local directory of wav files + transcripts (no download)')
    parser.add_argument('--logs_dir', type=str, default=None, help='root of the run directories (default ./logs)')
    parser.add_argument('--silent', action='store_true')
    return parser
