"""Drop-in runner: the orchestration of the reference's ``src/run_attack.py`` (epoch loop, clean / perturbed
evaluation, best tracking :151-167, StepLR :170-178, early stopping :181-183, final test + results.json :187-243, failure
report :265-279) with its eight defects fixed, on the HIP path.

    python -m paa_amd.run_attack --optimizer_type pgd --norm_type snr --snr_db 40 [--model_path DIR] [--data_dir DIR]
"""
from __future__ import annotations

import os
import sys

import torch

from .core import iso
from .training_utils import build, evaluation, parser, save, scoring_helpers, train


def main(args) -> int:
    if not torch.cuda.is_available():
        raise SystemExit("paa_amd.run_attack needs a GPU; there is no CPU fallback")
    if not str(args.device).startswith("cuda"):
        args.device = "cuda"
    # launched by torch.distributed.run with several ranks: one process per GPU, RCCL ("nccl") over xGMI, utterances
    # sharded over ranks; every rank keeps the same p, rank 0 writes the files
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        args.device = f"cuda:{local}"
        if not torch.distributed.is_initialized():
            backend = os.environ.get("PAA_DIST_BACKEND", "nccl")
            if backend == "nccl":
                torch.distributed.init_process_group("nccl", device_id=torch.device(args.device))
            else:
                torch.distributed.init_process_group(backend)
        if rank != 0:
            args.silent = True
    logger, start_epoch = build.create_logger(args)
    logger.info("Using device: %s", args.device)
    interp = iso.build_weight_interpolator()
    spl_thresh = build.init_phon_threshold_tensor(args)
    train_loader, eval_loader, test_loader, audio_len = build.create_data_loaders(args, rank, world)
    model, processor = build.load_model(args, max_batch=int(args.batch_size), length=audio_len)
    first = train_loader[0][0] if train_loader else None
    if world > 1 and first is not None:
        # the initial projection (build.py:301-304) sees the first GLOBAL batch, as the one-process run does: the shards are
        # gathered once at set-up (sizes may differ by a clip, hence the object collective)
        shards = [None] * world
        torch.distributed.all_gather_object(shards, first.cpu())
        first = torch.cat(shards, dim=0)
    p = build.init_perturbation(args, audio_len, spl_thresh, interp, first.to(args.device) if first is not None else None)
    if world > 1:          # identical by construction; the broadcast only guards against a rank-dependent resume file
        torch.distributed.broadcast(p.data, src=0)
    writer = rank == 0
    optimizer, scheduler = (build.create_optimizer(args, p) if args.optimizer_type == "adam" else (None, None))
    hist = {k: [] for k in ("train_ctc", "train_wer", "clean_ctc", "clean_wer", "pert_ctc", "pert_wer")}
    goal = scoring_helpers.Objective(args.attack_mode)      # targeted: perturbed WER down; untargeted: perturbed CTC up
    best_epoch, no_improve, best_eval = -1, 0, goal.worst
    pert_path = os.path.join(args.save_dir, "perturbation.pt")
    try:
        for epoch in range(start_epoch, int(args.num_epochs)):
            res = train.train_epoch(args=args, train_data_loader=train_loader, p=p, model=model, epoch=epoch,
                                    processor=processor, interp=interp, wer_metric=None, spl_thresh=spl_thresh,
                                    optimizer=optimizer)
            p = res.p
            hist["train_ctc"].append(res.avg_ctc); hist["train_wer"].append(res.avg_wer)
            clean = evaluation.evaluate(args, eval_loader, 0, model, processor, None, perturbed=False, epoch_number=epoch)
            pert = evaluation.evaluate(args, eval_loader, p, model, processor, None, perturbed=True, epoch_number=epoch)
            hist["clean_ctc"].append(clean.ctc); hist["clean_wer"].append(clean.wer)
            hist["pert_ctc"].append(pert.ctc); hist["pert_wer"].append(pert.wer)
            logger.info("[%d/%d] train ctc %.4f wer %.4f | eval clean ctc %.4f wer %.4f | eval perturbed ctc %.4f wer %.4f",
                        epoch + 1, args.num_epochs, res.avg_ctc, res.avg_wer, clean.ctc, clean.wer, pert.ctc, pert.wer)
            if writer:
                save.save_json_results(
                    save_dir=args.save_dir, norm_type=args.norm_type, attack_size=args.attack_size_string, epoch=epoch,
                    finished_training=False, eval_score_clean={"ctc": clean.ctc, "wer": clean.wer},
                    eval_score_perturbed={"ctc": goal.best(hist["pert_ctc"]), "wer": goal.best(hist["pert_wer"])},
                    train_score={"ctc": goal.best(hist["train_ctc"]), "wer": goal.best(hist["train_wer"])})
            current = pert.wer if args.attack_mode == "targeted" else pert.ctc     # run_attack.py:153
            if goal.improves(current, best_eval):
                no_improve, best_eval, best_epoch = 0, current, epoch
                if writer:
                    save.save_pert(p, pert_path)
                    save.save_by_epoch(args, p)
            else:
                no_improve += 1
            if scheduler is not None:
                scheduler.step()
                logger.info("[Epoch %d] new LR(s): %s", epoch, ", ".join(f"{lr:.6f}" for lr in scheduler.get_last_lr()))
            if no_improve >= int(args.early_stopping):
                logger.info("No improvements in %d epochs. Stopping early.", no_improve)
                break
        if world > 1:
            torch.distributed.barrier()          # rank 0's last perturbation.pt is on disk
        if os.path.exists(pert_path):
            p = save.load_pert(pert_path, args.device).to(args.device)
        pert_test = evaluation.evaluate(args, test_loader, p, model, processor, None, perturbed=True)
        clean_test = evaluation.evaluate(args, test_loader, 0, model, processor, None, perturbed=False)
        if writer:
            save.save_json_results(
                save_dir=args.save_dir, epoch=best_epoch, finished_training=True, norm_type=args.norm_type,
                attack_size=args.attack_size_string,
                best_train_score={"ctc": goal.best(hist["train_ctc"]), "wer": goal.best(hist["train_wer"])},
                eval_score_clean={"ctc": clean_test.ctc, "wer": clean_test.wer},
                eval_score_perturbed={"ctc": pert_test.ctc, "wer": pert_test.wer},
                final_test_clean={"ctc": clean_test.ctc, "wer": clean_test.wer},
                final_test_perturbed={"ctc": pert_test.ctc, "wer": pert_test.wer}, best_epoch=best_epoch)
        logger.info("done: best epoch %d | test clean ctc %.4f wer %.4f | test perturbed ctc %.4f wer %.4f", best_epoch,
                    clean_test.ctc, clean_test.wer, pert_test.ctc, pert_test.wer)
        return 0
    except Exception as e:      # noqa: BLE001  (run_attack.py:265-279: still leave a failure report behind)
        logger.exception("Run failed with an exception: %s", e)
        try:
            if rank == 0:
                save.save_json_results(save_dir=args.save_dir, epoch=-1, finished_training=False, norm_type=args.norm_type,
                                   attack_size=args.attack_size_string, error=str(e))
        except Exception:       # noqa: BLE001
            pass
        return 1


if __name__ == "__main__":
    sys.exit(main(parser.create_arg_parser().parse_args()))
