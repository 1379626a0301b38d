"""Child process of tests/test_gpu_sharded.py: one rank of play_games_sharded on the (shared) GPU.

    python tests/_sharded_worker.py RANK WORLD PORT OUT_DIR N_GAMES N_ITER MODE [BACKEND]

Started as a fresh process (nothing here runs before the interpreter starts: no GPU state is
inherited).  BACKEND gloo (default): the records are staged through the host, everything else is the
product path (real sessions, device packing, both collectives, the merge).  BACKEND nccl = RCCL, the
backend of the real 8-GPU job; one rank per device (so world size 1 on a one-GPU box)."""
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out_dir, n_games, n_iter, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    backend = sys.argv[8] if len(sys.argv) > 8 else "gloo"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    import torch
    import torch.distributed as dist

    from c4a0_amd import GameMetadata
    from c4a0_amd.distributed import play_games_sharded
    from tests.helpers import GraphSafeHashEval, hash_eval_torch

    if backend == "nccl":
        dev_index = rank % torch.cuda.device_count()
        torch.cuda.set_device(dev_index)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        device = f"cuda:{dev_index}"
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        device = "cuda:0"
    reqs = [GameMetadata(1000 + 7 * i, 0, 0) for i in range(n_games)]
    stats = {}
    extra = {}
    if mode == "rccl_live":
        # a training loop's use of the entry point: the process group has just run collectives on this device when the
        # HIP graphs are captured (broadcast of every weight tensor, barrier), a collective follows, and the call is
        # repeated in the same process -- RCCL's watchdog thread is alive (and polls events) during both captures
        from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
        torch.manual_seed(1337)
        net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 2, 2)), torch.device(device), dtype=torch.bfloat16)
        n_coll = 0
        for t in [net.tw0, net.tw, net.tbias, net.merged_w1, net.merged_b1] + net.pol_w + net.val_w:
            dist.broadcast(t, src=0)
            n_coll += 1
        dist.barrier()
        n_coll += 1
        kw = dict(evaluator=net, device=device, resident_games=16, concurrent_sessions=2)
        res = play_games_sharded(reqs, 64, n_iter, 6.6, 0.01, stats=stats, **kw)
        probe = torch.ones(1, device=device)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        second = play_games_sharded(reqs, 64, n_iter, 6.6, 0.01, **kw)
        extra = {"cbor_second_call": second.to_cbor(), "collectives_before_play": n_coll}
    elif mode == "net":        # the real bf16 4-block / 32-channel network (same weights on every rank: same seed)
        from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
        torch.manual_seed(1337)
        net = InferenceNet(ConnectFourNet(ModelConfig(4, 32, 4, 2)), torch.device(device), dtype=torch.bfloat16)
        res = play_games_sharded(reqs, 4096, n_iter, 6.6, 0.01, evaluator=net, device=device, resident_games=256, stats=stats)
    elif mode == "reclaim":    # n_mcts_iterations beyond the never-reclaimed arena's limit, halves as tight as the library takes, a look every 2nd launch
        period, n = 2, n_iter
        half = n + 2 + 8 + 2 * (2 * period * 2 + 16)
        res = play_games_sharded(reqs, 64, n_iter, 6.6, 0.01, evaluator=GraphSafeHashEval(), device=device, resident_games=4,
                                 concurrent_sessions=2, reclaim=True, reclaim_period=period, blocks_per_slot=2 * half, stats=stats)
        extra = {"reclaim_passes": stats["reclaim_passes"], "reclaim_blocks": stats["reclaim_blocks"]}
    elif mode == "graph2":     # HIP-graph replay, two concurrent sessions per rank
        res = play_games_sharded(reqs, 64, n_iter, 6.6, 0.01, evaluator=GraphSafeHashEval(), device=device,
                                 resident_games=16, concurrent_sessions=2, stats=stats)
    else:                    # eager single session per rank, slots refilled from the rank's queue
        res = play_games_sharded(reqs, 64, n_iter, 6.6, 0.01, evaluator=hash_eval_torch, device=device,
                                 resident_games=8, stats=stats)
    with open(os.path.join(out_dir, f"rank{rank}.pkl"), "wb") as f:
        pickle.dump({"cbor": res.to_cbor(), "allgather": stats["sample_allgather"], "games_done": stats.get("games_done"),
                     "host_loop": stats.get("host_loop"), **extra}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
