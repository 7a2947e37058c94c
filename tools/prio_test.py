import torch
print("priority range", torch.cuda.Stream.priority_range())
for p in (-1, 0, 1, 2):
    try:
        s = torch.cuda.Stream(priority=p); print("priority", p, "->", s.priority)
    except Exception as e:
        print("priority", p, "error", e)
