from .actor_critic import ActorCritic
